import ctypes, os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from cdv_slam_amd import synth, _lib, ops
lib = _lib.load(); dev = torch.device("cuda:0")
st = synth.make_state("global", features=False, frames=300, M=96, buffer_size=316, ht=384, wd=512)
T = lambda a: torch.as_tensor(a, device=dev)
poses0, patches0 = T(st.poses).float(), T(st.patches).float()
args = (T(st.intrinsics).float(), T(st.target).float(), T(st.weight).float(), torch.tensor([st.lmbda], device=dev), T(st.ii), T(st.jj), T(st.kk), st.cfg.M, st.t0, st.n, 1, True)
g = ops.GraphIndex(dev, E_cap=st.E, k_range=316 * 96); U = len(np.unique(st.kk))
for _ in range(3): ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g)
torch.cuda.synchronize()
buf = torch.zeros((4096, 16), dtype=torch.int64, device=dev)
fn = lib.cdv_set_stamps_ba; fn.argtypes = [ctypes.c_void_p]; assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
ops.ba_forward(poses0.clone(), patches0.clone(), *args, U_max=U, graph=g); torch.cuda.synchronize()
b = buf.cpu().numpy().astype(np.float64)
nb = 29
# one wave per block `me`: row 2000 + 64 me + kb holds, for the fold of block kb > me, {0: step start, 1: x_kb in hand, 2: folded};
# row 2000 + 64 me + me holds {0: own solve starts, 1: published}; row 2000 + 64 me + 63 holds {0: wave start}
row = lambda me, kb: b[2000 + 64 * me + kb]
# (cycle stamps -- s_memtime -- are per XCD: only differences inside one wave mean anything; across waves the 100 MHz real-time
# stamps 5 (published / wave start) and 6 (x in hand) are compared)
beg = np.array([row(me, me)[0] for me in range(nb)])
end = np.array([row(me, me)[1] for me in range(nb)])
print("per block, inside its wave (cycles):")
print("  own solve (chain + publish)        median %7.0f" % np.median(end - beg))
print("  fold of the block before it        median %7.0f" % np.median([row(me, me + 1)[2] - row(me, me + 1)[1] for me in range(nb - 1)]))
print("  folded -> own solve starts         median %7.0f" % np.median([beg[me] - row(me, me + 1)[2] for me in range(nb - 1)]))
pub = np.array([row(me, me)[5] for me in range(nb)])
start = min(row(me, 63)[5] for me in range(nb))
seen = np.array([row(me, me + 1)[6] - pub[me + 1] for me in range(nb - 1)])
d = pub[:-1] - pub[1:]
print("across waves (us): launch's first wave -> x_0 published %.2f" % ((pub[0] - start) / 100))
print("  x_k+1 published -> x_k published   median %.2f min %.2f max %.2f" % (np.median(d) / 100, d.min() / 100, d.max() / 100))
print("  published -> in the next block's hand  median %.2f min %.2f max %.2f" % (np.median(seen) / 100, seen.min() / 100, seen.max() / 100))
