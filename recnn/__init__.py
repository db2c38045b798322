"""`import recnn` resolves here when this repository (not the reference) is on sys.path: the name is handed over to
`recnn_amd`, the MI355X-native implementation of the same API, so that notebooks written against awarebayes/RecNN run without
a single changed line (`recnn.nn.DDPG`, `recnn.data.env.FrameEnv`, `recnn.utils.soft_update`, ... are recnn_amd's objects)."""
import recnn_amd as _impl

_impl.install_as(__name__)      # sys.modules["recnn"] (and every "recnn.<sub>") now IS recnn_amd: the import returns that module
