"""`models.models` module surface of the reference (models/models.py), served by ddnerf_amd."""
from ddnerf_amd.models import DDNerfModel, GeneralMipNerfModel  # noqa: F401
