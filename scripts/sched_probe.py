"""Block-schedule experiment for the N x N x N SYMM: time per variant (XCD remap on/off, GROUP_M).  Run under
rocprofv3 --pmc FETCH_SIZE to get memory-side traffic per dispatch (dispatch order = the order printed)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import mcgra_loader
pkg = mcgra_loader.load()
from mc_gra_amd import engine as E
from mc_gra_amd._lib import lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
scheds = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2 * 4, 2 * 16, 2 * 32, 2 * 16 + 1, 2 * 79]
torch.manual_seed(0)
A = torch.rand(n, 64, device="cuda") - 0.5
S = E.ssyrk_lower(A)                                   # symmetric, lower tile storage
B = torch.rand(n, n, device="cuda") - 0.5
C = torch.empty(n, n, device="cuda")
nbufs = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [2]
for sc, nb in [(a, b) for a in scheds for b in nbufs]:
    lib.mcgra_set_gemm_variant((nb & 255) | (sc << 8) | ((nb >> 8) << 16))
    E.ssymm_lower(S, B, out=C); torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); E.ssymm_lower(S, B, out=C); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"nbuf={nb} sched={sc:3d} (xcd_remap={'off' if sc & 1 else 'on'}, GROUP_M={(sc >> 1) or 8}): {min(ts)*1e3:7.3f} ms  {2*n**3/min(ts)/1e12:6.1f} TF", flush=True)
