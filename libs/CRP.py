"""Drop-in for the reference's libs/CRP.py: same module path, same class name,
MI355X underneath (bnpc_amd.model).  The reference driver decides by
``model.__module__`` whether error rates are learned (libs/MCMC.py:206-209),
so the classes must be DEFINED at exactly this module path."""
from bnpc_amd.model import CRP as _DeviceCRP
from bnpc_amd.model import EPSILON, TMIN, TMAX, log_EPSILON  # noqa: F401


class CRP(_DeviceCRP):
    pass
