"""Randomised sweep of the surfaces either side of the step kernels against their oracles: the negative sampler (bit-exact vs
the C restatement: random populations, id offsets, non-contiguous entity lists, neighbour tables, filter on / off, ragged
batches), the evaluator's ranks, the k-NN refresh (exact top-k sets), the common-space and space-mapping steps.
python tools/fuzz_aux.py [cases] [seed]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from gpu_util import dev_i32
from multike_amd.base.alignment import alignment_counts
from multike_amd.base.batch import neighbour_table
from multike_amd.runner import SpaceMappingState, run_space_mapping_steps
from multike_amd.sampling import KGSide, KnownTripleSet, sample_negatives
from multike_amd.tables import EmbeddingTable, StepEngine
from oracle import c_oracle as co
from oracle import eval_oracle as eo
from oracle import multike_oracle as mo

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
fails = {}


def report(kind, c, desc, msg):
    fails[kind] = fails.get(kind, 0) + 1
    print(f"{kind} case {c}: {desc}: {msg}", flush=True)


# ---- sampler ------------------------------------------------------------------------------------------------------------------
for c in range(cases):
    N = int(rng.choice([1, 2, 5, 10, 25, 31, 32, 33, 64]))
    n = int(rng.integers(N + 1, 4000)) if rng.random() < 0.8 else N + int(rng.integers(0, 3))
    lo = int(rng.integers(0, 100_000))
    contiguous = bool(rng.random() < 0.6)
    ents = np.arange(lo, lo + n) if contiguous else np.sort(rng.choice(lo + 3 * n, n, replace=False))
    n_total = int(ents.max()) + 1
    n_rel, P = int(rng.integers(1, 50)), int(rng.integers(1, 700))
    tri = np.stack([rng.choice(ents, 4 * P), rng.integers(0, n_rel, 4 * P), rng.choice(ents, 4 * P)], 1).astype(np.int32)
    use_known, use_near = bool(rng.random() < 0.7), bool(rng.random() < 0.4)
    desc = f"N={N} n={n} lo={lo} contiguous={contiguous} P={P} known={use_known} near={use_near}"
    try:
        ks = KnownTripleSet(dev_i32(tri[:, 0]), dev_i32(tri[:, 1]), dev_i32(tri[:, 2])) if use_known else None
        side = KGSide(ents, ks)
        ct = cv = None
        if use_near:
            K = int(rng.integers(N, max(N + 1, min(n, 200)) + 1))
            if K > n:
                K = n
            if K >= N:
                ct = np.zeros((n_total, K), np.int32)
                ct[ents] = np.stack([rng.choice(ents, K, replace=False) for _ in range(n)])
                cv = np.zeros(n_total, np.uint8); cv[ents] = rng.random(n) < 0.7
                side.set_neighbours(dev_i32(ct), torch.as_tensor(cv, device="cuda"))
        p = tri[rng.integers(0, len(tri), P)]
        seed, stream, off = (int(rng.integers(0, 2**31)), int(rng.integers(0, 2**31))), int(rng.integers(0, 1000)), int(rng.integers(0, 10**6))
        got = [x.cpu().numpy() for x in sample_negatives(tuple(dev_i32(p[:, k]) for k in range(3)), side, N, seed=seed, stream_id=stream, pos_offset=off)]
        exp = co.neg_sample(p[:, 0], p[:, 1], p[:, 2], N, n, ent_lo=side.ent_lo, ent_list=None if contiguous else ents.astype(np.int32),
                            cand_table=ct, cand_valid=cv, known=co.TripleSet(tri[:, 0], tri[:, 1], tri[:, 2]) if use_known else None,
                            seed=seed, stream_id=stream, pos_offset=off)
        if not all(np.array_equal(a, b) for a, b in zip(got, exp)):
            report("SAMPLER", c, desc, f"{sum(int((a != b).sum()) for a, b in zip(got, exp))} elements differ")
    except Exception as ex:  # noqa: BLE001
        report("SAMPLER", c, desc, f"{type(ex).__name__}: {str(ex)[:200]}")
print(f"sampler: {cases - fails.get('SAMPLER', 0)} / {cases} bit-exact")

# ---- evaluator ----------------------------------------------------------------------------------------------------------------
for c in range(cases // 2):
    d = int(rng.integers(1, 321)); n1 = int(rng.integers(1, 2500)); n2 = n1 + int(rng.integers(0, 2500))
    e2 = rng.standard_normal((n2, d)).astype(np.float32); e1 = (0.6 * e2[:n1] + rng.standard_normal((n1, d))).astype(np.float32)
    if rng.random() < 0.3 and n2 > 4:          # exact duplicates of gold columns: ties
        k = rng.integers(0, n1, max(1, n1 // 10)); e2[(k + 1) % n2] = e2[k]
    desc = f"d={d} n1={n1} n2={n2}"
    try:
        greater, ties, best = alignment_counts(e1, e2)
        r64, _ = eo.ranks(e1.astype(np.float64), e2.astype(np.float64))
        n = mo.l2_normalize_rows
        s = n(e1.astype(np.float64)) @ n(e2.astype(np.float64)).T
        gold = s[np.arange(n1), np.arange(n1)][:, None]
        g_lo, g_hi = (s > gold + 2e-6).sum(1), (s > gold - 2e-6).sum(1) - 1          # band the fp32 similarities may land in
        g = greater.cpu().numpy()
        if not np.all((g >= g_lo) & (g <= g_hi)):
            report("EVAL", c, desc, f"{int(((g < g_lo) | (g > g_hi)).sum())} ranks outside the fp32 band")
        t = ties.cpu().numpy()
        if t.min() < 1 or np.any(g + t - 1 > g_hi):
            report("EVAL", c, desc, "tie counts outside the band")
    except Exception as ex:  # noqa: BLE001
        report("EVAL", c, desc, f"{type(ex).__name__}: {str(ex)[:200]}")
print(f"evaluator: {cases // 2 - fails.get('EVAL', 0)} / {cases // 2} within the fp32 band of the float64 ranks")

# ---- k-NN refresh ---------------------------------------------------------------------------------------------------------------
for c in range(cases // 4):
    d = int(rng.integers(2, 260)); n = int(rng.integers(2, 30_000) if rng.random() < 0.5 else rng.integers(2, 600))
    k = int(rng.integers(1, max(2, min(n, 1000))))
    g = torch.Generator(device="cuda"); g.manual_seed(int(rng.integers(0, 2**31)))
    cen = torch.randn(max(2, n // 300), d, device="cuda", generator=g)
    e = torch.nn.functional.normalize(cen[torch.randint(0, cen.shape[0], (n,), device="cuda", generator=g)] + 0.8 * torch.randn(n, d, device="cuda", generator=g), dim=1)
    step = int(rng.integers(1, 4)); ids = (np.arange(n) * step + int(rng.integers(0, 50)))
    n_total = int(ids.max()) + 1 + int(rng.integers(0, 20))
    desc = f"d={d} n={n} k={k} id step={step}"
    try:
        table, valid = neighbour_table(e, ids.tolist(), k, n_total)
        idt = torch.as_tensor(ids, device="cuda")
        if int(valid.sum()) != n or not bool(valid[idt].all()):
            report("KNN", c, desc, "valid flags")
        pos = torch.full((n_total,), -1, dtype=torch.int64, device="cuda"); pos[idt] = torch.arange(n, device="cuda")
        tol = 3e-6
        for lo in range(0, n, 8192):
            sim = e[lo:lo + 8192].double() @ e.double().t()
            kth = torch.topk(sim, k, dim=1).values[:, -1:]
            t = pos[table[idt[lo:lo + 8192]].long()]
            if int(t.min()) < 0:
                report("KNN", c, desc, "an id outside the entity list"); break
            ts = t.sort(dim=1).values
            got = sim.gather(1, t)
            if k > 1 and not bool((ts[:, 1:] != ts[:, :-1]).all()):
                report("KNN", c, desc, "duplicate columns in a row"); break
            if float((kth - got).max()) > tol or not torch.equal((got > kth + tol).sum(1), (sim > kth + tol).sum(1)):
                report("KNN", c, desc, "not the top-k set"); break
    except Exception as ex:  # noqa: BLE001
        report("KNN", c, desc, f"{type(ex).__name__}: {str(ex)[:200]}")
print(f"k-NN refresh: {cases // 4 - fails.get('KNN', 0)} / {cases // 4} exact top-k sets")

# ---- common-space step + space-mapping step ----------------------------------------------------------------------------------------
for c in range(cases // 2):
    d = int(rng.integers(2, 300)); n_ent = int(rng.integers(4, 3000)); B = int(rng.integers(1, min(n_ent, 900) + 1))
    t0 = [mo.xavier_truncated_normal((n_ent, d), rng) for _ in range(4)]
    lit = t0[3] / np.maximum(np.linalg.norm(t0[3], axis=1, keepdims=True), 1e-12)
    w_name, w_cv = float(rng.uniform(0.2, 2.0)), float(rng.uniform(0.5, 2.0))
    desc = f"d={d} n_ent={n_ent} B={B}"
    try:
        ent = EmbeddingTable(n_ent, d, "ent", values=t0[0]); rv = EmbeddingTable(n_ent, d, "rv", values=t0[1]); av = EmbeddingTable(n_ent, d, "av", values=t0[2])
        name = EmbeddingTable(n_ent, d, "name", normalize=False, trainable=False, values=lit)
        eng = StepEngine()
        T = [a.astype(np.float64) for a in t0[:3]]; A = [np.full_like(a, 0.1) for a in T]; l64 = lit.astype(np.float32).astype(np.float64)
        for s in range(2):
            idx = rng.choice(n_ent, B, replace=False).astype(np.int32); di = dev_i32(idx)
            got = float(eng.alignment_step([(ent, di, name, di, w_cv * w_name), (ent, di, rv, di, w_cv), (ent, di, av, di, w_cv)], "cross_name", 0.02))
            exp = mo.common_space_step_dense(T[0], l64, T[1], T[2], A[0], A[1], A[2], idx, 0.02, w_name, w_cv)
            if abs(got - exp) > 2e-5 * abs(exp):
                report("COMMON", c, desc, f"loss {got} vs {exp}")
        for tb, ref, nm in ((ent, T[0], "ent"), (rv, T[1], "rv"), (av, T[2], "av")):
            if not np.allclose(tb.raw().cpu().numpy(), ref, rtol=3e-4, atol=2e-6 + 3e-5 * np.abs(ref).max()):
                report("COMMON", c, desc, f"{nm} max diff {np.abs(tb.raw().cpu().numpy() - ref).max():.2e}")
    except Exception as ex:  # noqa: BLE001
        report("COMMON", c, desc, f"{type(ex).__name__}: {str(ex)[:200]}")
print(f"common-space step: {cases // 2 - fails.get('COMMON', 0)} / {cases // 2} agree with the float64 oracle")

for c in range(cases // 4):
    d = int(rng.integers(2, 89)); n_ent = int(rng.integers(4, 3000)); B = int(rng.integers(1, min(n_ent, 900) + 1)); steps = int(rng.integers(1, 4))   # native path: dim <= 88
    ent0 = mo.xavier_truncated_normal((n_ent, d), rng); v0 = [mo.xavier_truncated_normal((n_ent, d), rng) for _ in range(3)]
    lit = v0[0] / np.maximum(np.linalg.norm(v0[0], axis=1, keepdims=True), 1e-12)
    Ms = [np.linalg.qr(rng.standard_normal((d, d)))[0] + 0.05 * rng.standard_normal((d, d)) for _ in range(3)]
    ow = float(rng.choice([0.5, 2.0]))
    desc = f"d={d} n_ent={n_ent} B={B} steps={steps} orthogonal_weight={ow}"
    try:
        ent = EmbeddingTable(n_ent, d, "ent_embeds", True, values=ent0); name = EmbeddingTable(n_ent, d, "name", False, trainable=False, values=lit)
        rv = EmbeddingTable(n_ent, d, "rv", True, values=v0[1]); av = EmbeddingTable(n_ent, d, "av", True, values=v0[2])
        st = SpaceMappingState([torch.as_tensor(m, dtype=torch.float32) for m in Ms], "cuda")
        idx = np.stack([rng.choice(n_ent, size=B, replace=False) for _ in range(steps)]).astype(np.int32)
        ring = run_space_mapping_steps(st, ent, [name, rv, av], torch.as_tensor(idx.reshape(-1), device="cuda"), np.arange(steps + 1) * B, "shared_comb", 1, 0.01, ow)
        got = ring.sum(dim=(1, 2)).cpu().numpy()
        E = ent0.astype(np.float64); accE = np.full_like(E, 0.1)
        tabs = [(lit.astype(np.float32).astype(np.float64), False), (v0[1].astype(np.float64), True), (v0[2].astype(np.float64), True)]
        M64 = [m.astype(np.float32).astype(np.float64) for m in Ms]; accM = [np.full_like(m, 0.1) for m in M64]
        for s in range(steps):
            L = mo.space_mapping_step_dense(E, accE, tabs, M64, accM, idx[s], 0.01, ow)
            if abs(got[s] - L) > 5e-5 * abs(L):
                report("MAPPING", c, desc, f"step {s} loss {got[s]} vs {L}")
        if not np.allclose(ent.raw().cpu().numpy(), E, rtol=3e-4, atol=2e-6 + 3e-5 * np.abs(E).max()):
            # yardstick: the same oracle replayed in float32 (NumPy order) on the same batches — rows read through l2_normalize with
            # ||w||^2 << lr W make the Adagrad step expand rounding noise (EXPERIMENTS R4.2); a kernel fault would not show up in it
            E32 = ent0.astype(np.float32); accE32 = np.full_like(E32, 0.1)
            tabs32 = [(t.astype(np.float32), tr) for t, tr in ((lit, False), (v0[1], True), (v0[2], True))]
            M32 = [m.astype(np.float32) for m in Ms]; accM32 = [np.full_like(m, 0.1) for m in M32]
            for s in range(steps):
                mo.space_mapping_step_dense(E32, accE32, tabs32, M32, accM32, idx[s], np.float32(0.01), np.float32(ow))
            report("MAPPING", c, desc, f"ent max diff {np.abs(ent.raw().cpu().numpy() - E).max():.2e} (the float32 replay of the oracle: "
                                       f"{np.abs(E32.astype(np.float64) - E).max():.2e}; median row norm {np.median(np.linalg.norm(E, axis=1)):.4f})")
        for k in range(3):
            if not np.allclose(st.M[k].cpu().numpy(), M64[k], rtol=3e-4, atol=3e-6):
                report("MAPPING", c, desc, f"M{k} max diff {np.abs(st.M[k].cpu().numpy() - M64[k]).max():.2e}")
    except Exception as ex:  # noqa: BLE001
        report("MAPPING", c, desc, f"{type(ex).__name__}: {str(ex)[:200]}")
print(f"space-mapping step: {cases // 4 - fails.get('MAPPING', 0)} / {cases // 4} agree with the float64 oracle")
sys.exit(1 if fails else 0)
