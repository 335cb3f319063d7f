#!/usr/bin/env python3
"""Randomised sweep of the query loop (opt-in): random geometric graphs with random extra edge weights, synthetic voxel
caches of which a random share collides, lazy and eager solving with several landmark counts -- statuses and costs against
scipy's Dijkstra on the valid sub-graph (an independent implementation), returned paths checked edge by edge.

    python tests/fuzz_queries.py [n_cases] [seed]          (FUZZ_QUERIES_NQ=n queries per case, default 300)
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from scipy.sparse import coo_matrix
    from scipy.sparse.csgraph import dijkstra
    from scipy.spatial import cKDTree
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    irt = importlib.import_module("interactive-rate-tendons_amd")
    T = irt.tendon
    bad = 0
    for case in range(n_cases):
        rng = np.random.default_rng([seed, case])
        N = int(rng.integers(2, 6))
        robot = T.TendonRobot(tendons=[T.TendonSpecs(C=[float(k)], D=[0.01]) for k in range(N)], specs=T.BackboneSpecs())
        vox = irt.VoxelOctree(64)
        vox.set_xlim(-1, 1); vox.set_ylim(-1, 1); vox.set_zlim(-1, 1)
        nb = 16 ** 3
        blocks = np.zeros(nb, np.uint64)
        blocks[rng.random(nb) < rng.uniform(0.02, 0.3)] = np.uint64(0xFFFFFFFFFFFFFFFF)      # occupied blocks
        vox.blocks[...] = blocks.reshape(vox.blocks.shape)
        chk = irt.VoxelBackboneValidityChecker(robot, irt.VoxelEnvironment(), vox)
        chk.engine.set_grid(vox.Nx(), vox.limits(), vox.blocks)                             # raw grid: no dilation wanted here
        V = int(rng.choice([300, 3000, 20000]))
        k = int(rng.integers(3, 9))
        st = rng.uniform(0, 20, (V, N))
        if rng.random() < 0.5:                                                              # two far-apart clusters: unreachable pairs
            st[: V // 3] += 100.0
        _, nbr = cKDTree(st).query(st, k=k + 1)
        e = np.stack([np.repeat(np.arange(V), k), nbr[:, 1:].reshape(-1)], 1)
        e = np.unique(np.sort(e, axis=1), axis=0).astype(np.int32)
        d = np.linalg.norm(st[e[:, 0]] - st[e[:, 1]], axis=1)
        w = d * (1.0 + rng.choice([0.0, 0.0, 0.5, 1.0], len(e)))                           # weights >= distance: the heuristic stays admissible
        E = len(e)
        # one block per item; an item collides when its block is occupied
        vid, eid = rng.integers(0, nb, V).astype(np.uint32), rng.integers(0, nb, E).astype(np.uint32)
        one = np.uint64(1)
        vc = dict(offsets=np.arange(V + 1), block_ids=vid, masks=np.full(V, one))
        ec = dict(offsets=np.arange(E + 1), block_ids=eid, masks=np.full(E, one))
        v_ok, e_ok = blocks[vid] == 0, blocks[eid] == 0
        keep = e_ok & v_ok[e[:, 0]] & v_ok[e[:, 1]]
        G = coo_matrix((np.r_[w[keep], w[keep]], (np.r_[e[keep, 0], e[keep, 1]], np.r_[e[keep, 1], e[keep, 0]])), shape=(V, V)).tocsr()
        nq = int(os.environ.get("FUZZ_QUERIES_NQ", "300"))      # (512 or more: the default switches share the rounds with the device search)
        q = rng.integers(0, V, (nq, 2))
        D = dijkstra(G, directed=False, indices=np.unique(q[:, 0]))
        row = {s: i for i, s in enumerate(np.unique(q[:, 0]))}
        want_cost = np.array([D[row[s], g] for s, g in q])
        ok = True
        for nl, eager in ((0, False), (16, False), (16, True), (3, False)):
            prm = irt.VoxelCachedLazyPRM(chk, st, e, weights=w)
            prm.set_caches(vc, ec)
            prm.prepare(nl)
            if eager:
                prm.revalidate()
            out = prm.solveWithRoadmap(q[:, 0], q[:, 1])
            for i in range(nq):
                s, g = q[i]
                if not v_ok[s]:
                    good = out["status"][i] == 2
                elif not v_ok[g]:
                    good = out["status"][i] == 3
                elif not np.isfinite(want_cost[i]):
                    good = out["status"][i] == 1
                else:
                    p = out["paths"][i]
                    good = out["status"][i] == 0 and len(p) >= 1 and p[0] == s and p[-1] == g and abs(out["cost"][i] - want_cost[i]) <= 1e-9 * max(1.0, want_cost[i])
                    if good and len(p) > 1:
                        pe = np.sort(np.stack([p[:-1], p[1:]], 1), axis=1)
                        key = pe[:, 0].astype(np.int64) * V + pe[:, 1]
                        ek = e[:, 0].astype(np.int64) * V + e[:, 1]
                        pos = np.searchsorted(ek, key)
                        good = bool((pos < E).all() and (ek[np.minimum(pos, E - 1)] == key).all() and keep[np.minimum(pos, E - 1)].all())
                        good = good and abs(w[pos].sum() - out["cost"][i]) <= 1e-9 * max(1.0, want_cost[i])
                if not good:
                    ok = False
                    print("   query", i, (int(s), int(g)), "landmarks", nl, "eager", eager, "status", int(out["status"][i]), "cost", float(out["cost"][i]), "want", float(want_cost[i]))
                    break
        print("case %d: dims=%d V=%d E=%d k=%d invalid v %.2f e %.2f reachable %.2f  %s" % (
            case, N, V, E, k, 1 - v_ok.mean(), 1 - e_ok.mean(), float(np.isfinite(want_cost).mean()), "ok" if ok else "MISMATCH"), flush=True)
        bad += not ok
    print("mismatching cases:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
