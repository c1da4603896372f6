"""MI355X-native batch-SOM engine behind the XPySom class surface of jcfaracco/xpysom-dask."""
from .xpysom import XPySom

__all__ = ["XPySom"]
