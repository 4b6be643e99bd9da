"""Same surface as the reference's ``models/__init__.py`` (one line re-exporting the model and the loss)."""
from .Effi_MVS_plus import Effi_MVS_plus, mvs_loss  # noqa: F401
