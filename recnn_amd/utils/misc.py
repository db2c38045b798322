"""soft_update / write_losses / DummyWriter (reference: recnn/utils/misc.py:1-35)."""
import torch

from .. import _lib as L

__all__ = ["soft_update", "write_losses", "DummyWriter"]


def soft_update(net, target_net, soft_tau=1e-2):
    """target = target*(1-tau) + param*tau per parameter pair, in `.parameters()` zip order
    (recnn/utils/misc.py:1-5), computed by `recnn_soft_update_flat` (fp32, same operand order).

    Networks adopted by a fused engine (see recnn_amd.nn.fused) also get their compute-layout shadows
    refreshed; this function itself only needs the parameters to live on the GPU.
    """
    pairs = list(zip(target_net.parameters(), net.parameters()))
    if not pairs:
        return
    for tp, p in pairs:
        if tp.device.type != "cuda" or p.device.type != "cuda":
            raise L.RecnnHipError("recnn_amd.utils.soft_update: parameters must be on the GPU (no CPU fallback); "
                                  "call .to('cuda') on the networks first")
        if not (tp.data.is_contiguous() and p.data.is_contiguous()):
            raise L.RecnnHipError("soft_update: non-contiguous parameter")
    stream = L.current_stream()
    for tp, p in pairs:
        src = p.data if p.dtype == torch.float32 else p.data.float()
        L.call("recnn_soft_update_flat", L.ptr(tp.data), L.ptr(src), tp.numel(), float(soft_tau), stream)
        torch.autograd.graph.increment_version(tp)     # the kernel wrote tp behind autograd's back
    from ..nn import fused
    fused.notify_params_changed(target_net)


def write_losses(writer, loss_dict, kind="train"):
    """Scalar logging of one step (recnn/utils/misc.py:8-18): `<kind>/<key>` per loss, then writer.close()."""
    step = loss_dict["step"]
    for key, value in loss_dict.items():
        if key != "step":
            writer.add_scalar(kind + "/" + key, value, global_step=step)
    writer.close()


class DummyWriter:
    """No-op stand-in for a tensorboard SummaryWriter (recnn/utils/misc.py:21-35)."""

    def add_figure(self, *args, **kwargs):
        pass

    def add_histogram(self, *args, **kwargs):
        pass

    def add_scalar(self, *args, **kwargs):
        pass

    def add_scalars(self, *args, **kwargs):
        pass

    def close(self, *args, **kwargs):
        pass
