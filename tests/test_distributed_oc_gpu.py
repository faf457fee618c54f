"""The owner-computes sharded trainer with the HIP kernels (mke_oc.hip): one rank (no collectives: every entity is local),
and two ranks SHARING the one GPU the test boxes have (collectives staged through gloo, `OcHostStagedComm`, because RCCL
refuses two ranks on one device) — against the float64 dense oracle on the same global batches.  The exchange logic itself
(slots, codes, chunks, epoch boundary) is covered under gloo in tests/test_distributed_oc_cpu.py; what is left unexercised
is RCCL's own transport at G > 1."""
import os
import socket

import numpy as np
import pytest
import torch

from oracle import c_oracle as co
from oracle import multike_oracle as mo

pytestmark = pytest.mark.gpu

N_ENT, N_REL, DIM, B, NEG, SEED = 3000, 20, 75, 300, 8, 11


def _reference(world, steps, n_ent=N_ENT, dim=DIM, neg=NEG, b=B, zipf=0.0):
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED, zipf=zipf)
    rng = np.random.default_rng(SEED)
    e = mo.xavier_truncated_normal((n_ent, dim), rng).astype(np.float32).astype(np.float64)
    r = mo.xavier_truncated_normal((N_REL, dim), rng).astype(np.float32).astype(np.float64)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), b * world, neg, device="cpu", seed=SEED)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    losses = []
    for i in range(steps):
        s = i % bat.steps
        if s == 0 and i > 0:
            bat.shuffle()
        ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        mid = lo + int(bat.cnt1[s])
        parts = []
        for k, (a, c) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:c], pr[a:c], pt[a:c], neg, ehi - elo, ent_lo=elo, known=sets[k], seed=bat.rng_seed,
                                       stream_id=bat.rng_stream + k, pos_offset=a))
        nn = [np.concatenate([parts[0][j], parts[1][j]]) for j in range(3)]
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), nn, 0.02)
        losses.append(L)
    return e, r, losses, bat.steps


def _make(rank, world, comm=None, chunks=1, excl=True, n_ent=N_ENT, dim=DIM, neg=NEG, b=B, peer=False, zipf=0.0, hot_min=None, em=None):
    from multike_amd.distributed_oc import OwnerComputesTrainer
    from multike_amd.synthetic import SyntheticKGs
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED, zipf=zipf)
    rng = np.random.default_rng(SEED)
    ent0 = mo.xavier_truncated_normal((n_ent, dim), rng)
    rel0 = mo.xavier_truncated_normal((N_REL, dim), rng)
    cls = OwnerComputesTrainer
    if hot_min is not None:             # the hub-row threshold of the shard (references per global step)
        cls = type("T", (OwnerComputesTrainer,), {"HOT_MIN": float(hot_min)})
    return cls(kgs, ent0, rel0, b, neg, rank, world, seed=SEED, lr=0.02, comm=comm, chunks=chunks,
                                exclusive_rows=excl, peer_direct=peer, entity_major=em)


@pytest.mark.parametrize("em", [True, False])   # entity-major second pass (round 6, the default) / the atomics form of rounds 2-5
@pytest.mark.parametrize("chunks,excl,dim,neg", [(1, True, 75, 8), (2, True, 75, 25), (1, False, 75, 8), (1, True, 256, 64),
                                                  (3, True, 20, 1),
                                                  (1, True, 75, 0)])   # positives only: the shape of the cross-KG loops
def test_one_rank_equals_dense_oracle(chunks, excl, dim, neg, em):
    """G = 1: every vector / code / gradient slot is local; the kernels alone against the float64 dense oracle over the
    first epoch's steps (the oracle's CPU batcher and the device batcher draw different permutations at the epoch
    boundary; that boundary is covered by the next test)."""
    n_ent = 3000 if dim < 256 else 1200
    _, _, _, spe = _reference(1, 1, n_ent, dim, neg)
    steps = min(spe, 9)
    tr = _make(0, 1, chunks=chunks, excl=excl, n_ent=n_ent, dim=dim, neg=neg, em=em)
    assert tr.em == em and tr._native_loop()[0]
    tr.run(0, steps)                    # the native step loop (mke_oc_steps); the other tests of this file drive step() from Python
    e, r, losses, _ = _reference(1, steps, n_ent, dim, neg)
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses), rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert tr.scratch_clean()
    assert tr.stride == dim or float(tr.ent[:, dim:].abs().max()) == 0.0


@pytest.mark.parametrize("em", [True, False])
@pytest.mark.parametrize("excl,dim,neg", [(True, 75, 8), (True, 75, 25), (False, 75, 33), (True, 256, 64), (True, 20, 1), (True, 75, 0)])
def test_quarter_wave_score_kernel_equals_dense_oracle(excl, dim, neg, em):
    """k_oc_score_q (a quarter-wave per positive, four positives per wavefront: the multi-rank shapes' kernel) forced on at one rank
    — every negative owned, so the per-quarter chains are as long as they get; codes in chunks of 16 (neg 25, 33, 64 span several),
    ragged last wavefronts — against the same float64 dense oracle as the wavefront-per-positive kernel."""
    from multike_amd import _lib
    n_ent = 3000 if dim < 256 else 1200
    _, _, _, spe = _reference(1, 1, n_ent, dim, neg)
    steps = min(spe, 7)
    old = _lib.set_option("oc_score_quarter", 1)
    try:
        tr = _make(0, 1, excl=excl, n_ent=n_ent, dim=dim, neg=neg, em=em)
        for i in range(steps):
            tr.step(i)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("oc_score_quarter", old)
    e, r, losses, _ = _reference(1, steps, n_ent, dim, neg)
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses), rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert tr.scratch_clean()


@pytest.mark.parametrize("em", [True, False])
@pytest.mark.parametrize("quarter,dim,neg", [(0, 75, 8), (1, 75, 8), (0, 20, 25), (1, 20, 25)])
def test_positives_needing_both_vectors_equal_dense_oracle(quarter, dim, neg, em):
    """A 60-entity KG: the sampler's re-draw rounds (known-triple hits) are frequent, their coin falls on the other side for half
    of them, so a fifth of the positives need BOTH HR and RT on the wire while the rest need one (slot -1 for the other) — the
    three cases of the one-vector-per-positive exchange in one step, both score kernels, against the dense float64 oracle."""
    from multike_amd import _lib
    n_ent = 60
    _, _, _, spe = _reference(1, 1, n_ent, dim, neg)
    steps = min(spe, 6)
    old = _lib.set_option("oc_score_quarter", quarter)
    try:
        tr = _make(0, 1, n_ent=n_ent, dim=dim, neg=neg, em=em)
        vpp = tr.check()["vectors_per_positive"]
        assert 1.05 < vpp < 1.6, vpp
        slots = torch.stack([tr._slot[0][:tr._n_all], tr._slot[1][:tr._n_all]])
        assert bool(((slots >= 0).sum(0) >= 1).all()) and bool((slots < 0).any()) and bool(((slots >= 0).sum(0) == 2).any())
        for i in range(steps):
            tr.step(i)
        torch.cuda.synchronize()
    finally:
        _lib.set_option("oc_score_quarter", old)
    e, r, losses, _ = _reference(1, steps, n_ent, dim, neg)
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses), rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert tr.scratch_clean()


@pytest.mark.parametrize("quarter", [0, 1])
def test_hub_rows_of_the_shard_are_the_same_function(quarter):
    """Zipf(1.2) head / tail entities: a few rows are head or tail of dozens of positives of every step.  The trainer declares them
    (mke_oc_step.hot): mke_oc_apply and the positives' own terms add to private copies behind the shard's rows, the update launch
    adds the copies.  Same tables as the run without the declaration (HOT_MIN out of reach), copies all zero afterwards."""
    from multike_amd import _lib
    from multike_amd.distributed_oc import OwnerComputesTrainer
    from multike_amd.synthetic import SyntheticKGs
    n_ent, dim, neg, b = 4000, 75, 8, 512
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED, zipf=1.2)
    rng = np.random.default_rng(SEED)
    ent0, rel0 = mo.xavier_truncated_normal((n_ent, dim), rng), mo.xavier_truncated_normal((N_REL, dim), rng)
    old = _lib.set_option("oc_score_quarter", quarter)
    try:
        out = []
        for hot_min in (20.0, 1e9):
            class T(OwnerComputesTrainer):
                HOT_MIN = hot_min
            tr = T(kgs, ent0, rel0, b, neg, 0, 1, seed=SEED, lr=0.02, entity_major=False)
            assert (tr.n_hot > 0) == (hot_min < 1e9), tr.n_hot
            for i in range(min(tr.steps, 8)):
                tr.step(i)
            torch.cuda.synchronize()
            assert tr.scratch_clean()
            out.append((tr.epoch_loss(), tr.gather_entity_table().cpu().numpy(), tr.rel[:, :dim].cpu().numpy(), tr.n_hot))
    finally:
        _lib.set_option("oc_score_quarter", old)
    assert out[0][3] >= 3
    np.testing.assert_allclose(out[0][0], out[1][0], rtol=2e-6)
    np.testing.assert_allclose(out[0][1], out[1][1], rtol=2e-4, atol=1e-5)   # the same float32 sums in two orders
    np.testing.assert_allclose(out[0][2], out[1][2], rtol=2e-4, atol=1e-5)   # the same float32 sums in two orders


def test_entity_major_hub_entities_equal_dense_oracle():
    """Zipf(1.2) head / tail entities under the entity-major second pass: a few rows (and relation rows) have reference lists of
    dozens to hundreds of entries per step — walked in list order by their quarter-wave — against the float64 dense oracle."""
    n_ent, dim, neg, b = 4000, 75, 8, 512
    _, _, _, spe = _reference(1, 1, n_ent, dim, neg, b, zipf=1.2)
    steps = min(spe, 6)
    tr = _make(0, 1, n_ent=n_ent, dim=dim, neg=neg, b=b, zipf=1.2, em=True)
    longest = int((tr._em["off"][1:tr._em["row0_host"][-1] + 1] - tr._em["off"][:tr._em["row0_host"][-1]]).max())
    assert longest > 64, longest                   # lists that span several 32-reference segments:
    assert int(tr._em["long0_host"][-1]) > 0       # ... cut into work items of their own, partial sums, a combine launch
    for i in range(steps):
        tr.step(i)
    e, r, losses, _ = _reference(1, steps, n_ent, dim, neg, b, zipf=1.2)
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses), rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert tr.scratch_clean()


def test_entity_major_plan_grows_when_a_rank_owns_more_than_its_share():
    """The reference lists are sized for 1.25 / G of the epoch's references; a rank that owns more (skewed ownership) reports the
    count, and the plan is redone in line at the exact size before the epoch starts.  Forced here by planning with room for 100
    references: same results as the float64 dense oracle afterwards."""
    n_ent, dim, neg = 3000, 75, 8
    _, _, _, spe = _reference(1, 1, n_ent, dim, neg)
    steps = min(spe, 5)
    tr = _make(0, 1, n_ent=n_ent, dim=dim, neg=neg, em=True)
    full = tr._em["capacity"]
    tr._em_capacity = {0: 100, 1: 100}
    tr._plan_epoch()                                   # overflows, re-plans at the reported size
    assert 100 < tr._em["capacity"] <= full and tr._em["n_refs_host"] <= tr._em["capacity"]
    tr.run(0, steps)
    e, r, losses, _ = _reference(1, steps, n_ent, dim, neg)
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses), rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)


@pytest.mark.parametrize("chunks,quarter", [(1, 0), (2, 1)])
def test_entity_major_step_is_bit_reproducible(chunks, quarter):
    """The entity-major step has no atomics on table rows: every row's gradient is summed in its reference list's order (the
    epoch plan sorts by (step, row, positive, kind)), the relation rows' partial gradients likewise — two runs from the same
    state give the same BITS (tables, accumulators and the epoch loss), across an epoch boundary.  (The atomics form of rounds
    2-5 differs from run to run in the last bits.)"""
    from multike_amd import _lib
    old = _lib.set_option("oc_score_quarter", quarter)
    try:
        out = []
        for _ in range(2):
            tr = _make(0, 1, chunks=chunks, neg=25, em=True)
            for i in range(tr.steps + 2):
                tr.step(i)
            torch.cuda.synchronize()
            out.append((tr.ent.clone(), tr.ent_acc.clone(), tr.rel.clone(), tr.rel_acc.clone(), tr.loss_ring.clone()))
    finally:
        _lib.set_option("oc_score_quarter", old)
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("em,native", [(True, True), (True, False), (False, True)])
def test_plain_sgd_rule_of_the_step_kernels(em, native):
    """The step descriptors carry the update rule (mke_oc_step.optimizer): Adagrad in every product path, plain SGD
    (code/MultiKE_model.py:24, tf.train.GradientDescentOptimizer) for C-ABI callers — the second pass's and the update launch's
    other branch.  Against the float64 dense oracle's gradients applied with its SGD rule, zipf 1.0 (hub rows, long lists)."""
    from multike_amd import _lib
    from multike_amd.distributed_oc import OwnerComputesTrainer
    from multike_amd.sampling import KGSide, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    n_ent, dim, neg, zipf, lr = 2500, 75, 8, 1.0, 0.05
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED, zipf=zipf)
    rng = np.random.default_rng(SEED)
    ent0 = mo.xavier_truncated_normal((n_ent, dim), rng)
    rel0 = mo.xavier_truncated_normal((N_REL, dim), rng)
    cls = type("SgdTrainer", (OwnerComputesTrainer,), {"OPTIMIZER": _lib.OPT_SGD})
    tr = cls(kgs, ent0, rel0, B, neg, 0, 1, seed=SEED, lr=lr, entity_major=em)
    steps = min(tr.steps, 6)
    acc0 = tr.ent_acc.clone()
    if native:
        tr.run(0, steps)
    else:
        for i in range(steps):
            tr.step(i)
    torch.cuda.synchronize()
    # the same steps on the oracle: gradients w.r.t. the normalised tables, then w -= lr * J^T g on the touched rows
    e, r = ent0.astype(np.float32).astype(np.float64), rel0.astype(np.float32).astype(np.float64)
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], KGSide(kgs.entities(0), None, device="cpu"),
                          KGSide(kgs.entities(1), None, device="cpu"), B, neg, device="cpu", seed=SEED)
    sets = [co.TripleSet(t[:, 0], t[:, 1], t[:, 2]) for t in kgs.triples]
    ph, pr, pt = (x.numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
    total = 0.0
    for s in range(steps):
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        mid = lo + int(bat.cnt1[s])
        parts = []
        for k, (a, c) in enumerate(((lo, mid), (mid, hi))):
            elo, ehi = kgs.ent_range[k]
            parts.append(co.neg_sample(ph[a:c], pr[a:c], pt[a:c], neg, ehi - elo, ent_lo=elo, known=sets[k], seed=bat.rng_seed,
                                       stream_id=bat.rng_stream + k, pos_offset=a))
        nn = [np.concatenate([parts[0][j], parts[1][j]]) for j in range(3)]
        L, ge, gr = mo.relation_view_step_dense(e, r, None, None, (ph[lo:hi], pr[lo:hi], pt[lo:hi]), nn, lr, update=False)
        mo.rows_update_sparse(e, None, ge, lr, True, "SGD")
        mo.rows_update_sparse(r, None, gr, lr, True, "SGD")
        total += L
    np.testing.assert_allclose(tr.epoch_loss(), total, rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :dim].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert torch.equal(tr.ent_acc, acc0)            # SGD has no slot: the accumulators are not touched
    assert tr.scratch_clean()


@pytest.mark.parametrize("zipf", [0.0, 1.0])
def test_epoch_plan_with_64_bit_keys_gives_the_same_lists(zipf):
    """mke_oc_em_plan sorts (step, row) keys as 32-bit words when n_steps * (n_local + n_rel) < 2^32 and as 64-bit words otherwise
    — a KG of tens of millions of entities; nothing at the tests' sizes gets there.  Option "oc_em_keys64" takes the 64-bit
    instantiation at any size: the same reference lists, touched rows, work items and long-row tables bit for bit (a stable sort of the
    same keys), and the same tables after an epoch and two steps (zipf 1.0: hub rows, long lists, the combine launch)."""
    from multike_amd import _lib
    runs = []
    for k64 in (0, 1):
        old = _lib.set_option("oc_em_keys64", k64)
        try:
            tr = _make(0, 1, neg=25, em=True, zipf=zipf, hot_min=None)
            em = tr._em
            n = int(em["n_refs_host"])
            S1 = tr.steps + 1
            items = int(em["host"][S1:2 * S1][-1])
            lists = [em["refs"][:2 * n].clone(), em["item_row"][:items].clone(), em["item_off"][:items + 1].clone(), em["item_part"][:items].clone(),
                     em["host"].clone()]
            tr.run(0, tr.steps + 2)
            torch.cuda.synchronize()
            runs.append(lists + [tr.ent.clone(), tr.ent_acc.clone(), tr.rel.clone(), tr.loss_ring.clone()])
        finally:
            _lib.set_option("oc_em_keys64", old)
    assert runs[0][0].numel() > 1000
    for a, b in zip(*runs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("chunks", [1, 3])
def test_native_step_loop_is_the_python_loop_bit_for_bit(chunks):
    """mke_oc_steps enqueues exactly what the Python step loop enqueues (entity-major form: no atomics, so the two runs agree to
    the bit), across an epoch boundary, in runs of steps that start and end inside epochs."""
    out = []
    for native in (False, True):
        tr = _make(0, 1, chunks=chunks, neg=25, em=True)
        n = tr.steps + 3
        if native:
            tr.run(0, 2)
            tr.run(2, n - 2)
        else:
            for i in range(n):
                tr.step(i)
        torch.cuda.synchronize()
        out.append((tr.ent.clone(), tr.ent_acc.clone(), tr.rel.clone(), tr.rel_acc.clone(), tr.loss_ring.clone()))
    for a, b in zip(*out):
        assert torch.equal(a, b)


@pytest.mark.parametrize("chunks", [1, 2])
def test_one_rank_across_the_epoch_boundary_equals_single_table_path(chunks):
    """Same global steps as the single-table StepEngine path (same device batcher, same seed => same shuffle): losses and
    tables agree past the epoch boundary (shuffle + re-plan: new negatives, codes, slots)."""
    from multike_amd.sampling import KGSide, KnownTripleSet, RelationBatcher
    from multike_amd.synthetic import SyntheticKGs
    from multike_amd.tables import EmbeddingTable, StepEngine
    n_ent, d, N = 3000, DIM, 10
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=N_REL, seed=SEED)
    rng = np.random.default_rng(SEED)
    ent0 = mo.xavier_truncated_normal((n_ent, d), rng)
    rel0 = mo.xavier_truncated_normal((N_REL, d), rng)
    tr = _make(0, 1, chunks=chunks, neg=N)
    E = EmbeddingTable(n_ent, d, "e", values=ent0)
    R = EmbeddingTable(N_REL, d, "r", values=rel0)
    sides = []
    for k in (0, 1):
        t = torch.as_tensor(kgs.triples[k], device="cuda")
        sides.append(KGSide(kgs.entities(k), KnownTripleSet(t[:, 0].contiguous(), t[:, 1].contiguous(), t[:, 2].contiguous())))
    bat = RelationBatcher(kgs.triples[0], kgs.triples[1], sides[0], sides[1], B, N, seed=SEED)
    eng = StepEngine()
    tot = 0.0
    nsteps = bat.steps + 3
    for s in range(nsteps):
        tr.step(s)
        if s > 0 and s % bat.steps == 0:
            bat.shuffle()
        pos, neg = bat.batch(s % bat.steps)
        l = float(eng.relation_step(E, R, "relation", pos, neg, neg_per_pos=N, lr=0.02).sum())
        if s >= 3:                  # the trainer's loss ring holds the last `steps` global steps
            tot += l
    np.testing.assert_allclose(tr.epoch_loss(), tot, rtol=2e-6)
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), E.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(tr.rel[:, :d].cpu().numpy(), R.raw().cpu().numpy(), rtol=1e-4, atol=1e-6)


def _two_rank_worker(rank, world, port, ret, chunks, steps, peer=False, neg=NEG, em=None, native=False):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)   # `port`: a rendezvous FILE (no TCP port to collide on)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        torch.cuda.set_device(0)
        if native == "overlap":         # the reduce-scatter on the communication stream, the second pass's gv-free work items under it
            os.environ["MKE_OC_OVERLAP_RS"] = "1"
        tr = _make(rank, world, comm=OcHostStagedComm(), chunks=chunks, peer=peer, neg=neg, em=em)
        if native:                      # mke_oc_steps: the schedule (two streams when chunks > 1) enqueued from C++, collectives by callback
            assert tr._native_loop()[0]
            tr.run(0, steps)
        else:
            for i in range(steps):
                tr.step(i)
        full = tr.gather_entity_table().cpu().numpy()
        ok = tr.scratch_clean()
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM].cpu().numpy().copy(), loss, ok))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chunks,peer,neg,em,native", [(1, False, NEG, True, False), (1, False, 0, True, False),
                                                       (2, False, NEG, True, True), (3, False, 25, True, True),
                                                       (2, False, NEG, False, True),
                                                       (1, False, NEG, False, False), (1, True, NEG, False, False)])   # neg 0: positives only
def test_two_ranks_on_one_gpu_equal_dense_oracle(chunks, peer, neg, em, native):
    """world_size 2 with the HIP kernels: owner = id % 2; each rank scores, for all 600 positives of the global step, the
    negatives whose corrupt entity it owns; gradient vectors summed across ranks; relation gradient all-reduced."""
    import torch.multiprocessing as mp
    import tempfile
    port = tempfile.mktemp(prefix="mke_rdv_")   # rendezvous file (init_method="file://..."): no TCP port to collide on
    world, steps = 2, 7
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    # peer = True: no all-gather / reduce-scatter — each process maps the other's send block and gradient inbox (IPC) and the
    # score kernel reads / writes them directly
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, ret, chunks, steps, peer, neg, em, native)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, ok = ret.get(timeout=480)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    e, r, losses, spe = _reference(world, steps, neg=neg)
    assert steps <= spe and ok
    np.testing.assert_allclose(loss, sum(losses), rtol=2e-6)
    np.testing.assert_allclose(full, e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(rel, r, rtol=2e-4, atol=2e-6)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,chunks,em,native", [(8, 1, True, True), (8, 2, True, True), (8, 1, False, False), (5, 2, True, True),
                                                    (8, 1, True, "overlap")])
def test_eight_ranks_on_one_gpu_equal_dense_oracle(world, chunks, em, native):
    """world_size 8 — the size the multi-GPU bench runs at: owner = id & 7 (the shift / mask instantiation of the score kernels
    and of the plan walks), 2,400 positives per global step, each rank scoring the eighth of the negatives it owns, entity-major
    second pass + the native step loop (collectives by callback into the host-staged communicator), against the float64 dense
    oracle.  world 5: the division instantiation."""
    import torch.multiprocessing as mp
    import tempfile
    port = tempfile.mktemp(prefix="mke_rdv_")
    steps = 4
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, world, port, ret, chunks, steps, False, NEG, em, native)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss, ok = ret.get(timeout=800)
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    e, r, losses, spe = _reference(world, steps)
    assert steps <= spe and ok
    np.testing.assert_allclose(loss, sum(losses), rtol=2e-6)
    np.testing.assert_allclose(full, e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(rel, r, rtol=2e-4, atol=2e-6)


def test_one_rank_rccl_collectives_run():
    """1-rank RCCL group: the trainer's own communicator class (all_gather_into_tensor / reduce_scatter_tensor / all_reduce)
    is importable and the G = 1 short-cuts leave it untouched."""
    import torch.distributed as dist
    from multike_amd.distributed_oc import OcComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    import tempfile
    dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        cm = OcComm()
        a = torch.arange(8, dtype=torch.float32, device="cuda")
        o = torch.empty(8, device="cuda")
        cm.all_gather(o, a)
        assert torch.equal(o, a)
        cm.reduce_scatter(o, a)
        assert torch.equal(o, a)
        w = cm.all_gather(o, a, async_op=True)
        w.wait()
        cm.all_reduce(a)
        tr = _make(0, 1, comm=cm)
        tr.step(0)
        torch.cuda.synchronize()
    finally:
        dist.destroy_process_group()


def _rccl_native_worker(ret, chunks, em, overlap=False):
    """(spawned: MKE_OC_FORCE_COLLECTIVES is read when the trainer is built, and a process has one default process group)"""
    import tempfile
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MKE_OC_FORCE_COLLECTIVES"] = "1"
    if overlap:
        os.environ["MKE_OC_OVERLAP_RS"] = "1"
    dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        from multike_amd.distributed_oc import OcRcclComm
        _, _, _, spe = _reference(1, 1, neg=25)
        steps = min(spe, 6)
        tr = _make(0, 1, chunks=chunks, neg=25, em=em)
        ok, cs = tr._native_loop()
        kind = (type(tr.comm).__name__, bool(ok), None if cs is None else int(cs.kind), bool(tr.force_collectives))
        tr.run(0, steps)
        torch.cuda.synchronize()
        ret.put((kind, tr.epoch_loss(), tr.gather_entity_table().cpu().numpy(), tr.rel[:, :DIM].cpu().numpy().copy(), tr.scratch_clean(), steps))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("chunks,em,overlap", [(1, True, False), (2, True, False), (1, True, True)])
def test_native_step_loop_over_rccl_entry_points(chunks, em, overlap):
    """mke_oc_steps with `mke_oc_comm` of kind NCCL: the library calls ncclAllGather / ncclReduceScatter / ncclAllReduce through the
    addresses the host side hands over (a real one-rank RCCL communicator, the G > 1 path forced: all three collectives of every
    step are issued for real; chunks 2: on the second stream, ordered by the call's events; overlap: the reduce-scatter on the second
    stream with the second pass's gradient-vector-free work items under it) — against the float64 dense oracle."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    p = ctx.Process(target=_rccl_native_worker, args=(ret, chunks, em, overlap))
    p.start()
    kind, loss, full, rel, clean, steps = ret.get(timeout=500)
    p.join(120)
    assert p.exitcode == 0
    assert kind == ("OcRcclComm", True, 0, True), kind          # RCCL through ctypes, native loop usable, MKE_OC_COMM_NCCL, forced
    e, r, losses, _ = _reference(1, steps, neg=25)
    np.testing.assert_allclose(loss, sum(losses), rtol=2e-6)
    np.testing.assert_allclose(full, e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(rel, r, rtol=2e-4, atol=2e-6)
    assert clean


def _rccl_epoch_worker(ret, forced):
    import tempfile
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    if forced:
        os.environ["MKE_OC_FORCE_COLLECTIVES"] = "1"
    dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        tr = _make(0, 1, neg=25, em=True)          # prefetch on (the default): the next epoch's plan is staged while this one trains
        n = 2 * tr.steps + 3
        tr.run(0, n)
        torch.cuda.synchronize()
        ret.put((type(tr.comm).__name__, bool(tr.force_collectives), tr.ent.cpu().numpy(), tr.ent_acc.cpu().numpy(), tr.rel.cpu().numpy(),
                 tr.loss_ring.cpu().numpy(), n))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_rccl_step_loop_across_epoch_boundaries_with_prefetch():
    """Round-5 advice: the G > 1 path over a real RCCL communicator ACROSS epoch boundaries with the plan prefetched — two
    boundaries here: the step collectives issued by mke_oc_steps, the plans' code all-gather by the host side on the same
    communicator at its fixed point of the step sequence (at one rank the sampling / gather / lists split is taken only when the
    collectives are forced: the split plan, its events and the hand-over are exercised).  Every collective is the identity at one
    rank and the step has no atomics, so the tables, accumulators and losses must equal, BIT FOR BIT, a run without any collective."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = []
    for forced in (True, False):
        ret = ctx.Queue()
        p = ctx.Process(target=_rccl_epoch_worker, args=(ret, forced))
        p.start()
        out.append(ret.get(timeout=500))
        p.join(120)
        assert p.exitcode == 0
    assert out[0][0] == "OcRcclComm" and out[0][1] and not out[1][1] and out[0][6] == out[1][6]
    for a, b in zip(out[0][2:6], out[1][2:6]):
        assert np.array_equal(a, b)


# ----------------------------------------------------------------------------------------------------------------------
# cross-KG inference loops on the sharded tables: positives only, `random.sample` batches of a triple list, loss x 2
# ----------------------------------------------------------------------------------------------------------------------
CK_TRIPLES, CK_B = 1700, 600


def _ck_setup():
    rng = np.random.default_rng(SEED + 3)
    triples = np.stack([rng.integers(0, N_ENT, CK_TRIPLES), rng.integers(0, N_REL, CK_TRIPLES), rng.integers(0, N_ENT, CK_TRIPLES)],
                       1).astype(np.int32)
    return triples, mo.xavier_truncated_normal((N_ENT, DIM), rng), mo.xavier_truncated_normal((N_REL, DIM), rng)


def _ck_list(triples, weighted):
    if not weighted:
        return triples
    w = np.random.default_rng(SEED + 4).uniform(0.2, 1.0, len(triples))
    return [(int(h), int(r), int(t), float(x)) for (h, r, t), x in zip(triples, w)]


def _ck_trainer(rank, world, comm=None, weighted=False):
    from multike_amd.distributed_oc import OwnerComputesTrainer, TripleListBatcher
    triples, ent0, rel0 = _ck_setup()
    bat = TripleListBatcher(_ck_list(triples, weighted), CK_B, device="cuda", seed=SEED)
    return OwnerComputesTrainer(None, ent0, rel0, CK_B, 0, rank, world, seed=SEED, lr=0.02, comm=comm, batcher=bat, scale=2.0)


def _ck_reference(steps, weighted=False):
    """Dense float64 oracle on the same draws (the batcher's draws are a function of (seed, epoch): built again here)."""
    from multike_amd.distributed_oc import TripleListBatcher
    triples, e, r = _ck_setup()
    e, r = e.astype(np.float64), r.astype(np.float64)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = TripleListBatcher(_ck_list(triples, weighted), CK_B, device="cuda", seed=SEED)
    losses = []
    for i in range(steps):
        s = i % bat.steps
        if s == 0 and i > 0:
            bat.shuffle()
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        pos = tuple(x[lo:hi].cpu().numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        assert len({tuple(t) for t in np.stack(pos, 1)}) >= 1
        pw = bat.pos_w[lo:hi].cpu().numpy().astype(np.float64) if weighted else None
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, pos, None, 0.02, pos_w=pw, scale=2.0)
        losses.append(L)
    return e, r, losses, bat.steps


@pytest.mark.parametrize("weighted", [False, True])
def test_one_rank_cross_kg_positive_steps_equal_dense_oracle(weighted):
    steps = 5                                  # 3 steps per epoch: crosses the epoch boundary
    tr = _ck_trainer(0, 1, weighted=weighted)
    for i in range(steps):
        tr.step(i)
    e, r, losses, spe = _ck_reference(steps, weighted)
    assert spe == 3
    np.testing.assert_allclose(tr.epoch_loss(), sum(losses[2:]), rtol=2e-6)   # the ring keeps the last 3 steps
    np.testing.assert_allclose(tr.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(tr.rel[:, :DIM].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert tr.scratch_clean()


def _ck_two_rank_worker(rank, world, port, ret, steps, weighted):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        torch.cuda.set_device(0)
        tr = _ck_trainer(rank, world, comm=OcHostStagedComm(), weighted=weighted)
        for i in range(steps):
            tr.step(i)
        full = tr.gather_entity_table().cpu().numpy()
        loss = tr.epoch_loss()
        if rank == 0:
            ret.put((full, tr.rel[:, :DIM].cpu().numpy().copy(), loss))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("weighted", [False, True])
def test_two_ranks_cross_kg_positive_steps_equal_dense_oracle(weighted):
    """Two ranks sharing the GPU: heads / tails of the sampled cross-KG triples live on either rank; 2 vectors per positive
    each way, nothing else."""
    import torch.multiprocessing as mp
    import tempfile
    port = tempfile.mktemp(prefix="mke_rdv_")
    world, steps = 2, 5
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_ck_two_rank_worker, args=(r, world, port, ret, steps, weighted)) for r in range(world)]
    for p in procs:
        p.start()
    full, rel, loss = ret.get(timeout=480)
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    e, r, losses, _ = _ck_reference(steps, weighted)
    np.testing.assert_allclose(loss, sum(losses[2:]), rtol=2e-6)
    np.testing.assert_allclose(full, e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(rel, r, rtol=2e-4, atol=2e-6)


def test_relation_group_on_shared_tables_one_rank():
    """The relation view and the weighted cross-KG relation-inference loop as two trainers on ONE pair of tables (own Adagrad
    slots, own tag ranges): a few steps of the first, then an epoch of the second, against the dense oracle."""
    from multike_amd.distributed_oc import OwnerComputesTrainer, TripleListBatcher
    steps_a = 4
    a = _make(0, 1)
    triples, _, _ = _ck_setup()
    lst = _ck_list(triples, True)
    b = OwnerComputesTrainer(None, None, None, CK_B, 0, 0, 1, seed=SEED, lr=0.02, batcher=TripleListBatcher(lst, CK_B, device="cuda", seed=SEED),
                             scale=2.0, tables_of=a)
    assert b.ent is a.ent and b.rel is a.rel and b.ent_acc is not a.ent_acc
    for i in range(steps_a):
        a.step(i)
    la = a.epoch_loss()
    for i in range(b.steps):
        b.step(i)
    lb = b.epoch_loss()
    # reference: the relation view's steps (same dense oracle as above), then the list's steps with their own accumulators
    e, r, losses, _ = _reference(1, steps_a)
    np.testing.assert_allclose(la, sum(losses), rtol=2e-6)
    ae, ar = np.full_like(e, 0.1), np.full_like(r, 0.1)
    bat = TripleListBatcher(lst, CK_B, device="cuda", seed=SEED)
    tot = 0.0
    for s in range(bat.steps):
        lo, hi = int(bat.off[s]), int(bat.off[s + 1])
        pos = tuple(x[lo:hi].cpu().numpy() for x in (bat.pos_h, bat.pos_r, bat.pos_t))
        L, _, _ = mo.relation_view_step_dense(e, r, ae, ar, pos, None, 0.02, pos_w=bat.pos_w[lo:hi].cpu().numpy().astype(np.float64), scale=2.0)
        tot += L
    np.testing.assert_allclose(lb, tot, rtol=2e-6)
    np.testing.assert_allclose(a.gather_entity_table().cpu().numpy(), e, rtol=2e-4, atol=2e-6)
    np.testing.assert_allclose(a.rel[:, :DIM].cpu().numpy(), r, rtol=2e-4, atol=2e-6)
    assert a.scratch_clean()


# ----------------------------------------------------------------------------------------------------------------------
# BASELINE configs[4] at its per-GPU shape, FULL size, in the sharded (owner-computes) form
# ----------------------------------------------------------------------------------------------------------------------
C5 = dict(n_ent=2_000_000, n_rel=2000, dim=256, neg=64, batch=5000)


def _c5_ent0(n_ent, dim, seed):
    """The initial entity table, identical on every rank: truncated normal at the xavier scale, drawn in float32 blocks."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    sigma = float(np.sqrt(2.6 / (n_ent + dim)))
    return (torch.randn(n_ent, dim, generator=g).clamp_(-2, 2) * sigma).numpy()


def _c5_full_size(rank, world, comm, steps=2):
    """`steps` global steps of the owner-computes trainer at |E| = 2M, |R| = 2000, dim 256, 64 negatives, 5000 positives PER
    RANK, checked on this rank against the float64 C oracle run on the COMPACTED problem (the rows the steps touch, renumbered:
    untouched rows take no part and must stay bit-identical — asserted on the shard).  Every rank recomputes the oracle for the
    whole global step (the tables start from the same seed) and checks the rows IT owns; nothing but the loss crosses ranks."""
    from multike_amd.distributed_oc import OwnerComputesTrainer
    from multike_amd.synthetic import SyntheticKGs
    n_ent, n_rel, d, N, P = C5["n_ent"], C5["n_rel"], C5["dim"], C5["neg"], C5["batch"]
    kgs = SyntheticKGs(n_ent=n_ent, n_rel=n_rel, triples_per_entity=1.0, seed=5)
    ent0 = _c5_ent0(n_ent, d, 5)
    rel0 = mo.xavier_truncated_normal((n_rel, d), np.random.default_rng(6))
    tr = OwnerComputesTrainer(kgs, ent0, rel0, P, N, rank, world, seed=2, lr=0.001, comm=comm, chunks=1)
    shard0 = tr.ent.clone()
    batches = []
    for s in range(steps):                       # the global batches (the Philox stream is indexed by the epoch position)
        pos, neg = tr.bat.batch(s)
        batches.append((tuple(x.cpu().numpy() for x in pos), tuple(x.cpu().numpy() for x in neg)))
    assert len(batches[0][0][0]) == P * world and len(batches[0][1][0]) == P * world * N
    for s in range(steps):
        tr.step(s)
    loss = tr.epoch_loss()
    used = np.unique(np.concatenate([a for pos, neg in batches for a in (pos[0], pos[2], neg[0], neg[2])]))
    remap = np.full(n_ent, -1, dtype=np.int64)
    remap[used] = np.arange(len(used))
    e64 = ent0[used].astype(np.float64)
    r64 = rel0.astype(np.float64)
    a64, b64 = np.full_like(e64, 0.1), np.full_like(r64, 0.1)
    orc = co.RelationStepOracle(len(e64), n_rel, d, np.float64)
    exp = 0.0
    for pos, neg in batches:
        exp += orc.step(e64, r64, a64, b64, (remap[pos[0]], pos[1], remap[pos[2]]), (remap[neg[0]], neg[1], remap[neg[2]]), 0.001)
    np.testing.assert_allclose(loss, exp, rtol=2e-6)
    mine = used[used % world == rank]
    loc = torch.as_tensor(mine // world, device="cuda")
    np.testing.assert_allclose(tr.ent[loc][:, :d].cpu().numpy(), e64[remap[mine]], rtol=1e-4, atol=5e-7)
    np.testing.assert_allclose(tr.ent_acc[loc][:, :d].cpu().numpy(), a64[remap[mine]], rtol=1e-3, atol=1e-7)
    np.testing.assert_allclose(tr.rel[:, :d].cpu().numpy(), r64, rtol=1e-4, atol=5e-7)
    mask = torch.ones(tr.ent.shape[0], dtype=torch.bool, device="cuda")
    mask[loc] = False
    assert torch.equal(tr.ent[mask], shard0[mask])                       # the shard's untouched rows: bit-identical
    assert tr.scratch_clean()
    moved = float((tr.ent[loc] - shard0[loc]).abs().max())
    assert moved > 0.0
    return dict(rank=rank, used=int(len(used)), owned=int(len(mine)), loss=loss, oracle_loss=exp)


@pytest.mark.timeout(900)
def test_full_size_c5_owner_computes_one_rank_over_rccl():
    """configs[4] per-GPU shape at full size on the SHARDED path, G = 1, the communicator a real 1-rank RCCL group."""
    import tempfile
    import torch.distributed as dist
    from multike_amd.distributed_oc import OcComm
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("nccl", init_method="file://" + tempfile.mktemp(prefix="mke_rdv_"), rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        out = _c5_full_size(0, 1, OcComm())
        assert out["used"] > 500_000 and out["owned"] == out["used"]     # 2 steps x ~300K distinct rows of 2M
    finally:
        dist.destroy_process_group()
        torch.cuda.empty_cache()


def _c5_worker(rank, world, port, ret):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    dist.init_process_group("gloo", init_method=f"file://{port}", rank=rank, world_size=world)
    try:
        from multike_amd.distributed_oc import OcHostStagedComm
        torch.cuda.set_device(0)
        ret.put(_c5_full_size(rank, world, OcHostStagedComm()))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(1500)
def test_full_size_c5_owner_computes_two_ranks_on_one_gpu():
    """The same at world 2 (two ranks sharing the GPU, 1M-row shards, 10,000 positives / 650K scored triples per global step,
    collectives staged through gloo): each rank scores, for all positives, the negatives whose corrupt entity it owns; 2 x
    10.5K x 256 floats cross the ranks each way per step."""
    import tempfile
    import torch.multiprocessing as mp
    port = tempfile.mktemp(prefix="mke_rdv_")
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    procs = [ctx.Process(target=_c5_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    outs = [ret.get(timeout=1400) for _ in range(2)]
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(o["rank"] for o in outs) == [0, 1]
    assert outs[0]["used"] == outs[1]["used"] and outs[0]["owned"] + outs[1]["owned"] == outs[0]["used"]
    assert abs(outs[0]["loss"] - outs[1]["loss"]) <= 1e-9 * abs(outs[0]["loss"])
