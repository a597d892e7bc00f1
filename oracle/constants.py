"""TEST INFRASTRUCTURE - constants shared by the oracle modules
(libs/CRP.py:10-14)."""
import numpy as np

# The reference traps log(0) and invalid operations and uses the resulting
# FloatingPointError as control flow (libs/CRP.py:10, 94-98, 110-114, 769-773)
np.seterr(divide='raise', over='ignore', under='ignore', invalid='raise')

EPSILON = np.finfo(np.float64).resolution      # 1e-15
TMIN = 1e-5
TMAX = 1 - TMIN
log_EPSILON = np.log(EPSILON)
