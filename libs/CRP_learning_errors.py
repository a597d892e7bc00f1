"""Drop-in for the reference's libs/CRP_learning_errors.py (see libs/CRP.py)."""
from bnpc_amd.model import CRP_errors_learning as _DeviceCRPErrors
from libs.CRP import CRP  # noqa: F401  (the reference re-exports it too)


class CRP_errors_learning(_DeviceCRPErrors):
    pass
