import sys
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
import kpop_amd
from oracle import ca_ref
kpop_amd.init(0)
rng = np.random.RandomState(1)
I, J = 500, 40
col = rng.poisson(5.0, size=I).astype(np.float64) + 1
N = np.repeat(col[:, None], J, axis=1)
tw, inertia, T = kpop_amd.ca(N, True)
print("identical spectra: finite", np.isfinite(tw).all(), np.isfinite(T).all(), "inertia", inertia[:3], "max |twisted|", np.abs(tw).max())
col2 = rng.poisson(5.0, size=I).astype(np.float64) + 1
N2 = N.copy(); N2[:, J // 2:] = col2[:, None]
tw, inertia, T = kpop_amd.ca(N2, True)
tw_o, in_o, T_o = ca_ref.ca(N2, True)
print("two groups: inertia", inertia[:3], "ref", in_o[:3], "finite", np.isfinite(tw).all(), np.isfinite(T).all())
a = ca_ref.align_signs(tw, tw_o, axis=1)
print("   first dimension agrees:", np.allclose(a[:, 0], tw_o[:, 0], atol=1e-9 * np.abs(tw_o).max()))
