"""Loss bookkeeping / debug figures used by the training notebooks (reference: recnn/utils/plot.py).
Off the hot path: only `Plotter.log_losses` runs per step; figures need matplotlib (imported lazily)."""
import numpy as np
import torch

__all__ = ["Plotter", "pairwise_distances_fig", "pairwise_distances", "smooth", "smooth_gauss"]


def _pairwise(embs: torch.Tensor):
    x = embs.detach().float()
    euc = torch.cdist(x, x)
    xn = torch.nn.functional.normalize(x, dim=1)
    cos = 1.0 - xn @ xn.t()
    return cos.cpu().numpy(), euc.cpu().numpy()


def pairwise_distances_fig(embs):
    """Cosine / euclidean pairwise-distance heat maps of a few action vectors (plot.py:9-31)."""
    import matplotlib
    matplotlib.use("Agg", force=False)
    import matplotlib.pyplot as plt
    cos, euc = _pairwise(embs)
    fig = plt.figure(figsize=(16, 10))
    for pos, (mat, title) in enumerate(((cos, "Cosine"), (euc, "Euclidian"))):
        ax = fig.add_subplot(1, 2, pos + 1)
        fig.colorbar(ax.matshow(mat))
        ax.set_title(title)
        ax.axis("off")
    fig.suptitle("Action pairwise distances")
    plt.close()
    return fig


def pairwise_distances(embs):
    pairwise_distances_fig(embs).show()


def smooth(scalars, weight):
    """Exponential smoothing, weight in [0, 1) (plot.py:39-47)."""
    out, last = [], scalars[0]
    for v in scalars:
        last = last * weight + (1 - weight) * v
        out.append(last)
    return out


def smooth_gauss(arr, var):
    from scipy import ndimage
    return ndimage.gaussian_filter1d(arr, var)


class Plotter:
    """Keeps the `loss_layout` lists of an Algo up to date and plots them (plot.py:54-93)."""

    def __init__(self, loss, style):
        self.loss = loss
        self.style = style
        self.smoothing = lambda x: smooth_gauss(x, 4)

    def set_smoothing_func(self, f):
        self.smoothing = f

    def log_loss(self, key, item, test=False):
        self.loss["test" if test else "train"][key].append(item)

    def log_losses(self, losses, test=False):
        for key, val in losses.items():
            self.log_loss(key, val, test)

    def plot_loss(self):
        import matplotlib.pyplot as plt
        for row in self.style:
            fig, axes = plt.subplots(1, len(row), figsize=(16, 6))
            axes = np.atleast_1d(axes)
            for ax, key in zip(axes, row):
                ax.set_title(key)
                ax.plot(self.loss["train"]["step"], self.smoothing(self.loss["train"][key]), "b-", label="train")
                ax.plot(self.loss["test"]["step"], self.loss["test"][key], "r-.", label="test")
            plt.legend()
        plt.show()
