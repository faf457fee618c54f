"""CPU compute backend for the distributed host logic, built on the oracle (TEST infrastructure only).
It lets `ShardedRelationTrainer`'s exchange code run under gloo without a GPU; the product backend is
multike_amd.distributed.HipBackend."""
import numpy as np

from oracle import c_oracle as co
from oracle import multike_oracle as mo


class OracleBackend:
    device_type = "cpu"

    def make_known(self, h, r, t):
        return co.TripleSet(h.numpy(), r.numpy(), t.numpy())

    def sample(self, pos, pos_offset, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        ph, pr, pt = (x.numpy() for x in pos)
        kg = pos_kg.numpy()
        n1 = int((kg == 0).sum())
        assert np.all(kg[:n1] == 0) and np.all(kg[n1:] == 1)  # [KG1 part | KG2 part]
        for k, (a, b, side) in enumerate(((0, n1, side1), (n1, len(kg), side2))):
            if b > a:
                assert side.ent_list is None
                nh, nr, nt = co.neg_sample(ph[a:b], pr[a:b], pt[a:b], neg_per_pos, side.n_ent, ent_lo=side.ent_lo,
                                           known=side.known, seed=seed, stream_id=stream_id + k, pos_offset=pos_offset + a)
                for o, v in zip(out, (nh, nr, nt)):
                    o.numpy()[a * neg_per_pos:b * neg_per_pos] = v

    def sample_at(self, pos, pos_index, pos_kg, side1, side2, neg_per_pos, seed, stream_id, out):
        """Explicit epoch positions: split into runs of consecutive positions (one per step slice) and sample each run
        with the offset-based oracle -- the RNG is indexed by epoch position, so this is what one launch draws."""
        import torch
        idx = pos_index.numpy()
        if len(idx) == 0:
            return
        kg = pos_kg.numpy().astype(np.int32)
        brk = (np.diff(idx) != 1) | (np.diff(kg) < 0)          # position gap, or a new [KG1 | KG2] step begins
        cuts = [0] + [int(k) + 1 for k in np.nonzero(brk)[0]] + [len(idx)]
        for a, b in zip(cuts[:-1], cuts[1:]):
            sub = tuple(torch.from_numpy(np.empty((b - a) * neg_per_pos, dtype=np.int32)) for _ in range(3))
            self.sample(tuple(x[a:b] for x in pos), int(idx[a]), pos_kg[a:b], side1, side2, neg_per_pos, seed, stream_id, sub)
            for o, v in zip(out, sub):
                o.numpy()[a * neg_per_pos:b * neg_per_pos] = v.numpy()

    # ---- bookkeeping ops (numpy restatements of mke_rowset_build / _remap / mke_rows_gather_padded / _scatter_add) ----
    def rowset_build(self, streams, flags, counts, req, id_map, overflow, n_ranks, capacity):
        ids = np.concatenate([x.numpy() for x in streams])
        f, cnt, rq, im = flags.numpy(), counts.numpy(), req.numpy(), id_map.numpy()
        for i in ids:
            if f[i]:
                continue
            f[i] = 1
            o = int(i) % n_ranks
            if cnt[o] < capacity:
                rq[o * capacity + cnt[o]] = int(i) // n_ranks
                im[i] = o * capacity + cnt[o]
            else:
                overflow.numpy()[0] = 1
                im[i] = n_ranks * capacity          # the pad row behind the compact row set
            cnt[o] += 1

    def rowset_remap(self, streams, outs, id_map, flags, reset_req=None, reset_counts=None, want=None, slot_of=None,
                     n_ranks=0, capacity=0):
        for ids, out in zip(streams, outs):
            i = ids.numpy()
            out.numpy()[:] = id_map.numpy()[i]
            flags.numpy()[i] = 0
        if reset_req is not None:
            reset_req.numpy()[:] = -1
        if reset_counts is not None:
            reset_counts.numpy()[:] = 0
        if want is not None:
            w, so = want.numpy(), slot_of.numpy()
            k = np.nonzero(w >= 0)[0]
            so[w[k].astype(np.int64) * n_ranks + k // capacity] = k % capacity

    def reduce_update(self, rel, shard, tag, dim, lr):
        """mke_rows_update_multi with a slot_of table: rank-ordered sum of the returned rows, one update per row."""
        table, acc, grad, touched, normalize = rel
        self.update(table, acc, grad, touched, tag, dim, normalize, lr)
        G, C = shard["n_ranks"], shard["capacity"]
        ent, so, src = shard["table"], shard["slot_of"].numpy(), shard["src_rows"].numpy()
        n = ent.shape[0]
        so2 = so[:n * G].reshape(n, G)
        g = np.zeros((n, src.shape[1]), dtype=src.dtype)
        for r in range(G):
            rows = np.nonzero(so2[:, r] >= 0)[0]
            g[rows] += src[r * C + so2[rows, r]]
        hit = (so2 >= 0).any(axis=1)
        so2[hit] = -1
        import torch
        touched = torch.from_numpy(np.where(hit, tag, 0).astype(np.int32))
        self.update(ent, shard["acc"], torch.from_numpy(g), touched, tag, dim, shard["normalize"], lr)

    def update_pair(self, t0, t1, tag, dim, lr):
        for (table, acc, grad, touched, normalize) in (t0, t1):
            self.update(table, acc, grad, touched, tag, dim, normalize, lr)

    def gather_padded(self, table, idx, out, zero_rows=None):
        i = idx.numpy()
        o = out.numpy()
        o[:] = 0
        o[i >= 0] = table.numpy()[i[i >= 0]]
        if zero_rows is not None:
            zero_rows.numpy()[:] = 0

    def scatter_add(self, idx, rows, dim, grad, touched, tag, reset_req=None, reset_counts=None):
        i = idx.numpy()
        m = i >= 0
        np.add.at(grad.numpy(), i[m], rows.numpy()[m])
        touched.numpy()[i[m]] = tag
        if reset_req is not None:
            reset_req.numpy()[:] = -1
        if reset_counts is not None:
            reset_counts.numpy()[:] = 0

    def score(self, ent, ent_norm, rel, rel_norm, dim, pos, neg, neg_per_pos, grad_ent, grad_rel, touched_ent, touched_rel,
              tag, loss_partials):
        p = tuple(x.numpy() for x in pos)
        n = tuple(x.numpy() for x in neg)
        L, ge, gr = mo.relation_view_step_dense(ent.numpy(), rel.numpy(), None, None, p, n, 0.0, ent_norm=ent_norm,
                                                rel_norm=rel_norm, update=False)
        grad_ent.numpy()[...] += ge
        grad_rel.numpy()[...] += gr
        te, tr = touched_ent.numpy(), touched_rel.numpy()
        for a in (p[0], p[2], n[0], n[2]):
            te[a] = tag
        tr[p[1]] = tag
        tr[n[1]] = tag
        lp = loss_partials.numpy()
        lp[:] = 0
        lp[0] = L

    def update(self, table, acc, grad, touched, tag, dim, normalize, lr):
        w, a, g = table.numpy(), acc.numpy(), grad.numpy()
        rows = np.arange(len(w)) if touched is None else np.nonzero(touched.numpy() == tag)[0]
        gg = mo.l2_normalize_rows_backward(w[rows], g[rows]) if normalize else g[rows]
        a[rows] += gg * gg
        w[rows] -= lr * gg / np.sqrt(a[rows])
        g[rows] = 0


class OcOracleBackend(OracleBackend):
    """NumPy restatement of the owner-computes kernels (mke_oc.hip: pack_codes / bases / count / score / apply) and of the
    row update, float64 — the CPU backend of `multike_amd.distributed_oc.OwnerComputesTrainer` under gloo.  Same block
    layout as the device: [capacity] HR vectors | [capacity] RT vectors, rows of `stride` elements."""

    EPS = 1e-12

    def block_elems(self, capacity, stride):
        return 2 * capacity * stride

    def pack_codes(self, pos_h, neg_h, neg_t, neg_per_pos, codes):
        ph = np.repeat(pos_h.numpy(), neg_per_pos)
        nh, nt = neg_h.numpy(), neg_t.numpy()
        cd = np.where(nh != ph, (nh << 1) | 1, nt << 1).astype(np.int64).reshape(-1, neg_per_pos)
        any_h = (cd & 1).any(axis=1)
        any_t = ((cd & 1) == 0).any(axis=1)
        cd[:, 0] |= np.where(any_h, 0x80000000, 0) | np.where(any_t | ~any_h, 0x40000000, 0)     # the group's need flags
        codes.numpy()[:] = cd.reshape(-1).astype(np.uint32).view(np.int32)

    @classmethod
    def _nrm(cls, x):
        return x / np.sqrt(np.maximum((x * x).sum(-1, keepdims=True), cls.EPS))

    def bases(self, tr, st, send):
        G, C, S, d = tr.world, tr.C, tr.stride, tr.dim
        ent, rel = tr.ent.numpy()[:, :d], tr.rel.numpy()[:, :d]
        out = send.numpy().reshape(2 * C, S)
        ph, pr, pt = st.pos_h.numpy(), st.pos_r.numpy(), st.pos_t.numpy()
        oh, ot = st.own_h.numpy(), st.own_t.numpy()
        out[:len(oh), :d] = self._nrm(ent[ph[oh] // G]) + self._nrm(rel[pr[oh]])            # needed vectors only (own lists)
        out[C:C + len(ot), :d] = self._nrm(rel[pr[ot]]) - self._nrm(ent[pt[ot] // G])

    def _codes_of(self, tr, st):
        """[n_pos, N] codes of the part, assembled from every home rank's run."""
        n, N = st.pos_h.numel(), tr.N
        cd = st.codes.numpy()
        out = np.zeros((n, N), dtype=np.int64)
        for g, off in enumerate(st.code_off):
            a, e = g * st.per, min(n, (g + 1) * st.per)
            if e > a:
                out[a:e] = cd[off:off + (e - a) * N].reshape(e - a, N)
        return out & 0x3FFFFFFF      # without the group flags

    def count(self, tr, st):
        G = tr.world
        rc = tr.ref_count.numpy()
        c = self._codes_of(tr, st).reshape(-1) >> 1
        np.add.at(rc, c[c % G == tr.rank] // G, 1)
        for ids in (st.pos_h.numpy(), st.pos_t.numpy()):          # every owned head / tail of the step's positives
            np.add.at(rc, ids[ids % G == tr.rank] // G, 1)

    def _row_update(self, w, a, g, lr):
        """Jacobian of the normalisation + Adagrad on one raw row (in place)."""
        s = float((w * w).sum())
        inv = 1.0 / np.sqrt(max(s, self.EPS))
        gg = (g - w * ((w * g).sum() * inv * inv if s > self.EPS else 0.0)) * inv
        a += gg * gg
        w -= lr * gg / np.sqrt(a)

    def score(self, tr, st, v_all, g_all, loss_partials):
        G, C, S, d, N, rank = tr.world, tr.C, tr.stride, tr.dim, tr.N, tr.rank
        V = v_all.numpy().reshape(G, 2 * C, S)
        Gout = g_all.numpy().reshape(G, 2 * C, S)
        ent, acc, eg = tr.ent.numpy(), tr.ent_acc.numpy(), tr.ent_grad.numpy()
        rel, rg = tr.rel.numpy(), tr.rel_grad.numpy()
        te, trl = tr.ent_touched.numpy(), tr.rel_touched.numpy()
        rc = tr.ref_count.numpy() if tr.ref_count is not None else None
        ph, pr, pt = st.pos_h.numpy(), st.pos_r.numpy(), st.pos_t.numpy()
        sh, stt = st.slot_h.numpy(), st.slot_t.numpy()
        codes = self._codes_of(tr, st)
        loss = 0.0
        sc = float(getattr(tr, "scale", 1.0))
        for i in range(len(ph)):
            HR = V[ph[i] % G, sh[i], :d] if sh[i] >= 0 else None        # slot -1: that vector does not travel
            RT = V[pt[i] % G, C + stt[i], :d] if stt[i] >= 0 else None
            gHR, gRT = np.zeros(d), np.zeros(d)
            # the positive's own term: the owner of t against HR (d = HR - t^), or the owner of h against RT (d = h^ + RT)
            other = pt[i] if HR is not None else ph[i]
            if other % G == rank:
                row = other // G
                eh = self._nrm(ent[row, :d])
                dv = HR - eh if HR is not None else eh + RT
                x = float(dv @ dv)
                pw = float(st.pos_w[i]) if st.pos_w is not None else 1.0
                loss += pw * np.log1p(np.exp(x))
                g = pw * sc * 2.0 / (1.0 + np.exp(-x)) * dv
                if HR is not None:
                    gHR += g
                    eg[row, :d] -= g
                else:
                    gRT += g
                    eg[row, :d] += g
                te[row] = st.tag
            for cd in codes[i]:
                c, head = int(cd) >> 1, int(cd) & 1
                if c % G != rank:
                    continue
                row = c // G
                ch = self._nrm(ent[row, :d])
                dv = ch + RT if head else HR - ch
                y = float(dv @ dv)
                loss += np.log1p(np.exp(-y))
                g = sc * -2.0 / (1.0 + np.exp(y)) * dv
                if head:
                    gRT += g
                else:
                    gHR += g
                gc = g if head else -g
                if rc is not None and rc[row] == 1:
                    self._row_update(ent[row, :d], acc[row, :d], gc, tr.lr)
                    rc[row] = 0
                else:
                    eg[row, :d] += gc
                    te[row] = st.tag
            if sh[i] >= 0:
                Gout[ph[i] % G, sh[i], :] = 0
                Gout[ph[i] % G, sh[i], :d] = gHR
            if stt[i] >= 0:
                Gout[pt[i] % G, C + stt[i], :] = 0
                Gout[pt[i] % G, C + stt[i], :d] = gRT
        lp = loss_partials.numpy()
        lp[:] = 0
        lp[0] = loss * sc

    def apply(self, tr, st, gv):
        G, C, S, d = tr.world, tr.C, tr.stride, tr.dim
        g = gv.numpy().reshape(2 * C, S)
        eg, rg = tr.ent_grad.numpy(), tr.rel_grad.numpy()
        te, trl = tr.ent_touched.numpy(), tr.rel_touched.numpy()
        ph, pr, pt = st.pos_h.numpy(), st.pos_r.numpy(), st.pos_t.numpy()
        for k, pos in enumerate(st.own_h.numpy()):
            eg[ph[pos] // G, :d] += g[k, :d]
            rg[pr[pos], :d] += g[k, :d]
            te[ph[pos] // G] = trl[pr[pos]] = st.tag
        for k, pos in enumerate(st.own_t.numpy()):
            eg[pt[pos] // G, :d] -= g[C + k, :d]
            rg[pr[pos], :d] += g[C + k, :d]
            te[pt[pos] // G] = trl[pr[pos]] = st.tag

    def update(self, tr, tag):
        d = tr.dim
        # relation table: every row (after the all-reduce a row may carry a gradient no local triple touched)
        for w, a, g, t in ((tr.rel, tr.rel_acc, tr.rel_grad, None), (tr.ent, tr.ent_acc, tr.ent_grad, tr.ent_touched)):
            wn, an, gn = w.numpy(), a.numpy(), g.numpy()
            for row in (range(len(wn)) if t is None else np.nonzero(t.numpy() == tag)[0]):
                self._row_update(wn[row, :d], an[row, :d], gn[row, :d].copy(), tr.lr)
                gn[row] = 0
                if w is tr.ent and tr.ref_count is not None:
                    tr.ref_count.numpy()[row] = 0

    def run(self, tr, k, tag, phases, c, loss_slot):
        from multike_amd.distributed_oc import APPLY, BASES, COUNT, SCORE, UPDATE
        st = tr._part_step(k, tag)
        if phases & BASES:
            self.bases(tr, st, tr._send[c])
        if phases & COUNT and tr.ref_count is not None:
            self.count(tr, st)
        if phases & SCORE:
            self.score(tr, st, tr._v_all[c], tr._g_all[c], tr.loss_ring[loss_slot])
        if phases & APPLY:
            self.apply(tr, st, tr._gv[c])
        if phases & UPDATE:
            self.update(tr, tag)


# ---- NumPy backends of multike_amd/distributed_views.py (float64) ------------------------------------------------------
class OracleAttrBackend:
    device_type = "cpu"

    def __init__(self, view, ent_shard, attr0, lit, cnn_params, tables_of=None):
        from oracle import attr_cnn_oracle as ao
        self.ao = ao
        if tables_of is not None:   # another graph on the same tables: own CNN set, own accumulators
            self.ent, self.attr, self.lit = tables_of.ent, tables_of.attr, tables_of.lit
        else:
            self.ent = np.array(ent_shard, dtype=np.float64)
            self.attr, self.lit = np.array(attr0, dtype=np.float64), np.array(lit, dtype=np.float64)
        self.p = {k: np.array(v, dtype=np.float64) for k, v in cnn_params.items()}
        self.acc_p = {k: np.full_like(v, 0.1) for k, v in self.p.items()}
        self.acc_ent, self.acc_attr = np.full_like(self.ent, 0.1), np.full_like(self.attr, 0.1)
        self.loss = 0.0

    def forward(self, view, lh, ia, iv, w, scale):
        import torch
        self.lh, self.ia, self.w, self.scale = lh, np.asarray(ia, dtype=np.int64), w, scale
        self.hs = mo.l2_normalize_rows(self.ent)[lh] if len(self.ent) else np.zeros((0, view.dim))
        _, self.c = self.ao.forward(self.p, self.hs, self.attr[self.ia], self.lit[np.asarray(iv, dtype=np.int64)])
        return torch.tensor([self.c["S"]], dtype=torch.float64)

    def tail(self, view, S):
        import torch
        self.S = float(S)
        loss, self.g_h, self.g_out, T, self.t = self.ao.dp_tail(self.c, self.hs, self.w, self.scale, self.S)
        self.loss += loss
        return torch.tensor([T], dtype=torch.float64)

    def backward(self, view, T):
        import torch
        g = self.ao.dp_backward(self.p, self.c, self.t, self.g_out, self.S, float(T))
        self.ge = np.zeros_like(self.ent)
        np.add.at(self.ge, self.lh, self.g_h)
        self.ga = np.zeros_like(self.attr)
        np.add.at(self.ga, self.ia, g["as"])
        self.flat = np.concatenate([g[k].reshape(-1) for k in self.ao.PARAM_NAMES])
        return [torch.from_numpy(self.flat), torch.from_numpy(self.ga)]

    def update(self, view):
        mo.adagrad_dense(self.ent, self.acc_ent, mo.l2_normalize_rows_backward(self.ent, self.ge), view.lr)
        mo.adagrad_dense(self.attr, self.acc_attr, self.ga, view.lr)
        o = 0
        for k in self.ao.PARAM_NAMES:
            n = self.p[k].size
            mo.adagrad_dense(self.p[k], self.acc_p[k], self.flat[o:o + n].reshape(self.p[k].shape), view.lr)
            o += n

    # replicated-compute step (ShardedAttributeView mode="replicated"): the same three hooks as the HIP backend
    def gather_heads(self, view, pos, lh, n):
        import torch
        out = np.zeros((n, view.dim))
        out[pos] = self.ent[lh]
        return torch.from_numpy(out)

    def replicated_step(self, view, H, ia, iv, w, scale):
        import torch
        hs = mo.l2_normalize_rows(H.numpy())
        self.ia = np.asarray(ia, dtype=np.int64)
        _, c = self.ao.forward(self.p, hs, self.attr[self.ia], self.lit[np.asarray(iv, dtype=np.int64)])
        loss, self.g_h_all, g_out, T, t = self.ao.dp_tail(c, hs, w, scale, c["S"])
        if view.rank == 0:
            self.loss += loss
        g = self.ao.dp_backward(self.p, c, t, g_out, c["S"], T)
        ga = np.zeros_like(self.attr)
        np.add.at(ga, self.ia, g["as"])
        return torch.from_numpy(np.concatenate([np.concatenate([g[k].reshape(-1) for k in self.ao.PARAM_NAMES]), ga.reshape(-1)]))

    def apply_replicated(self, view, pos, lh, canon):
        canon = canon.numpy()
        npar = sum(self.p[k].size for k in self.ao.PARAM_NAMES)
        self.flat, self.ga = canon[:npar], canon[npar:].reshape(self.attr.shape)
        self.ge = np.zeros_like(self.ent)
        np.add.at(self.ge, lh, self.g_h_all[pos])
        self.update(view)

    def take_loss(self):
        import torch
        v = torch.tensor([self.loss], dtype=torch.float64)
        self.loss = 0.0
        return v

    def tables(self):
        return self.ent, self.attr, self.p


class OracleCommonSpaceBackend:
    device_type = "cpu"

    def __init__(self, view, shards):
        self.t = {k: np.array(v, dtype=np.float64) for k, v in shards.items()}
        self.acc = {k: np.full_like(self.t[k], 0.1) for k in ("ent", "rv", "av")}
        self.loss = 0.0

    def step(self, view, rows):
        if len(rows) == 0:
            return
        t, a = self.t, self.acc
        self.loss += mo.common_space_step_dense(t["ent"], t["name"], t["rv"], t["av"], a["ent"], a["rv"], a["av"], rows, view.lr,
                                                view.cv_name_weight, view.cv_weight)

    def take_loss(self):
        import torch
        v = torch.tensor([self.loss], dtype=torch.float64)
        self.loss = 0.0
        return v

    def tables(self):
        return {k: self.t[k] for k in ("ent", "rv", "av")}


class OracleSpaceMappingBackend:
    """NumPy (float64) part-wise evaluation of the space-mapping step (oracle.space_mapping_grads cut at the two batch-wide
    sums), for multike_amd.distributed_views.ShardedSpaceMapping under gloo."""
    device_type = "cpu"

    def __init__(self, view, ent0, views0, matrices):
        self.ent = np.array(ent0, dtype=np.float64)
        self.acc_ent = np.full_like(self.ent, 0.1)
        self.views = [np.array(v, dtype=np.float64) for v in views0]
        self.M = [np.array(m, dtype=np.float64) for m in matrices]
        self.accM = [np.full_like(m, 0.1) for m in self.M]
        self.loss = np.zeros(2)

    def forward(self, view, rows):
        import torch
        self.rows = np.asarray(rows, dtype=np.int64)
        E = mo.l2_normalize_rows(self.ent) if len(self.ent) else self.ent
        self.F = E[self.rows]
        self.V = [(mo.l2_normalize_rows(t) if len(t) else t)[self.rows] for t in self.views]
        self.P = [v @ m for v, m in zip(self.V, self.M)]
        return torch.tensor([float(np.sum(p * p)) for p in self.P], dtype=torch.float64)

    def tail(self, view, S):
        import torch
        S = S.numpy()
        self.inv = [1.0 / np.sqrt(max(s, mo.L2_EPS)) for s in S]
        self.S = S
        self.out = [p * i for p, i in zip(self.P, self.inv)]
        self.gF = np.zeros_like(self.F)
        self.G = []
        T = []
        for out in self.out:
            diff = self.F - out
            self.loss[0] += float(np.sum(diff * diff))
            self.gF += 2.0 * diff
            self.G.append(-2.0 * diff)
            T.append(float(np.sum(self.G[-1] * out)))
        return torch.tensor(T, dtype=torch.float64)

    def backward(self, view, T):
        import torch
        T = T.numpy()
        g = []
        for k in range(len(self.M)):
            dP = self.inv[k] * (self.G[k] - self.out[k] * T[k]) if self.S[k] > mo.L2_EPS else self.inv[k] * self.G[k]
            g.append(self.V[k].T @ dP)
        self.gM = torch.as_tensor(np.stack(g))
        return self.gM

    def update(self, view):
        gM = self.gM.numpy()
        d = self.M[0].shape[0]
        for k, M in enumerate(self.M):
            Q = M @ M.T - np.eye(d)
            self.loss[1] += view.orthogonal_weight * float(np.sum(Q * Q)) + view.norm_w * float(np.sum(M * M))
            g = gM[k] + view.orthogonal_weight * 4.0 * (Q @ M) + view.norm_w * 2.0 * M
            mo.adagrad_dense(M, self.accM[k], g, view.lr)
        if len(self.rows):
            ge = np.zeros_like(self.ent)
            np.add.at(ge, self.rows, self.gF)
            mo.adagrad_dense(self.ent, self.acc_ent, mo.l2_normalize_rows_backward(self.ent, ge), view.lr)

    def take_loss(self):
        import torch
        v = torch.as_tensor(self.loss.copy())
        self.loss[:] = 0.0
        return v

    def tables(self):
        return self.ent, np.stack(self.M)


class OracleAutoEncoderBackend:
    """NumPy (float64) part-wise evaluation of one auto-encoder step (oracle.literal_oracle.loss_and_grads cut at the two
    batch-wide sums), for multike_amd.distributed_views.ShardedAutoEncoder under gloo."""
    device_type = "cpu"

    def __init__(self, view, params):
        from oracle import literal_oracle as lo
        self.lo = lo
        self.p = {k: np.array(v, dtype=np.float64) for k, v in params.items()}
        self.acc = {k: np.full_like(v, 0.1) for k, v in self.p.items()}
        self.n = len(view.dims) - 1
        self.keys = sorted(self.p)
        self.loss = np.zeros(1)

    def encode(self, view, x_rows, m_global):
        import torch
        lo, p = self.lo, self.p
        self.x, self.mg = np.asarray(x_rows, dtype=np.float64).reshape(-1, view.dims[0]), int(m_global)
        self.acts = [self.x]
        h = self.x
        for i in range(self.n):
            h = lo._act(h @ p[f"encoder_h{i}"] + p[f"encoder_b{i}"], view.active)
            self.acts.append(h)
        return torch.tensor([float(np.sum(h * h))], dtype=torch.float64)

    def decode(self, view, S):
        import torch
        lo, p = self.lo, self.p
        code = self.acts[-1]
        self.S = float(S[0])
        self.inv = 1.0 / np.sqrt(max(self.S, lo.L2_EPS)) if view.normalize else 1.0
        h = code * self.inv
        self.dacts = [h]
        for i in range(self.n):
            h = lo._act(h @ p[f"decoder_h{i}"] + p[f"decoder_b{i}"], view.active)
            self.dacts.append(h)
        diff = h - self.x
        denom = self.mg * view.dims[0]
        self.loss[0] += float(np.sum(diff * diff)) / denom
        self.g = {}
        d = 2.0 * diff / denom
        for i in reversed(range(self.n)):
            d = d * lo._act_grad(self.dacts[i + 1], view.active)
            self.g[f"decoder_h{i}"] = self.dacts[i].T @ d
            self.g[f"decoder_b{i}"] = d.sum(0)
            d = d @ p[f"decoder_h{i}"].T
        self.dcn = d
        return torch.tensor([float(np.sum(d * code))], dtype=torch.float64)

    def backward(self, view, T):
        import torch
        lo, p = self.lo, self.p
        code, d = self.acts[-1], self.dcn
        if view.normalize:
            d = self.inv * d - code * (self.inv ** 3) * float(T[0]) if self.S > lo.L2_EPS else self.inv * d
        for i in reversed(range(self.n)):
            d = d * lo._act_grad(self.acts[i + 1], view.active)
            self.g[f"encoder_h{i}"] = self.acts[i].T @ d
            self.g[f"encoder_b{i}"] = d.sum(0)
            d = d @ p[f"encoder_h{i}"].T
        self.flat = torch.as_tensor(np.concatenate([self.g[k].ravel() for k in self.keys]))
        return self.flat

    def update(self, view):
        flat, o = self.flat.numpy(), 0
        g = {}
        for k in self.keys:
            g[k] = flat[o:o + self.p[k].size].reshape(self.p[k].shape)
            o += self.p[k].size
        self.lo.adagrad_step(self.p, self.acc, g, view.lr)

    def take_loss(self):
        import torch
        v = torch.as_tensor(self.loss.copy())
        self.loss[:] = 0.0
        return v

    def params(self):
        return {k: v.copy() for k, v in self.p.items()}
