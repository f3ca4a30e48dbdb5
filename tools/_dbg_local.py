import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pycollo_amd import problems
from pycollo_amd.engine import NlpEngine
from pycollo_amd.sharding import LocalRoot, LocalShard, global_tile_plan
name, K, order, world = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
prob = problems.REGISTRY[name](K=K, order=order)
eng = NlpEngine(prob, device=0)
rng = np.random.default_rng(4)
x = rng.uniform(-0.45, 0.45, eng.num_x); lam = rng.normal(size=eng.num_c)
c, G, H = (a.copy() for a in eng.evaluate_all(x, 0.9, lam))
ref = np.concatenate([c, G, H])
root = LocalRoot(eng, world); plan = root.plan
dev = torch.device("cuda", 0)
dx, dl = torch.from_numpy(x).to(dev), torch.from_numpy(lam).to(dev)
tp = global_tile_plan(eng.model, eng.meshes, device=0)
s = torch.cuda.Stream(device=dev)
with torch.cuda.stream(s):
    for r in range(world):
        ls = LocalShard(eng.model, r, world, device=0, meshes=eng.meshes, plan=tp)
        ls.set_scaling(eng.V_ocp, eng.r_ocp, eng.W_ocp, 1.0)
        packed = ls.evaluate_packed(dx, dl, s.cuda_stream); s.synchronize()
        p = packed.cpu().numpy(); idx = plan.index[r]
        keep = idx < len(ref)
        bad = np.nonzero(p[keep] != ref[idx[keep]])[0]
        print("rank", r, "ranges", ls.ranges, "halo", ls.halo, "tiles", [ls.engine.phase_tiles(i)[0][:4].tolist() for i in range(len(ls.ranges))], "orders", [ls.engine.phase_tile_orders(i)[:4].tolist() for i in range(len(ls.ranges))], "info", {k: ls.engine.info[k] for k in ("threads_per_block", "waves_per_tile", "n_tiles_total")})
        print("   mismatches", len(bad), "of", int(keep.sum()), "first", bad[:6], "global pos", idx[keep][bad[:6]], "got", p[keep][bad[:3]], "want", ref[idx[keep]][bad[:3]])
        # locate in local segments
        if len(bad):
            o = 0
            for (a, b), (ga, gb) in zip(ls.segments, plan.segments[r]):
                n = b - a
                inseg = bad[(bad >= o) & (bad < o + n)]
                if len(inseg): print("     segment local", (a, b), "global", (ga, gb), "bad", len(inseg), "of", n, "first offs", (inseg[:5] - o).tolist())
                o += n
        ls.close()
# partial sums: the global handle's own (two-launch form) against the ranks'
oG, oH = plan.num_c, plan.num_c + plan.nnz_G
buf = torch.zeros(plan.total, dtype=torch.float64, device=dev)
for ip, ((k0, nred), off) in enumerate(zip(plan.tiles, plan.part_off)):
    if nred: eng.set_partials_buffer(ip, buf[off:off + (len(k0) - 1) * nred])
with torch.cuda.stream(s):
    eng.launch_bulk_only(dx, dl, buf[:oG], buf[oG:oH], buf[oH:oH + plan.nnz_H], s.cuda_stream); s.synchronize()
refp = buf.cpu().numpy()
with torch.cuda.stream(s):
    for r in range(world):
        ls = LocalShard(eng.model, r, world, device=0, meshes=eng.meshes, plan=tp)
        ls.set_scaling(eng.V_ocp, eng.r_ocp, eng.W_ocp, 1.0)
        packed = ls.evaluate_packed(dx, dl, s.cuda_stream); s.synchronize()
        p = packed.cpu().numpy(); idx = plan.index[r]
        keep = idx >= len(ref)
        bad = np.nonzero(p[keep] != refp[idx[keep]])[0]
        print("rank", r, "partials: mismatches", len(bad), "of", int(keep.sum()), "first", bad[:5], "got", p[keep][bad[:3]], "want", refp[idx[keep]][bad[:3]], "local part_off", ls.part_off, "nred", [ls.engine.phase_tiles(i)[1] for i in range(len(ls.ranges))])
        ls.close()
