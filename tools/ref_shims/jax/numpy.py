import numpy as _np

ndarray = _np.ndarray
