"""CPU model of ONE RANK of the sharded BPE trainer -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

It does, in plain Python over a list of ids, what one rank's kernels do between the collectives of
ecg_byte_amd/trainer.py::bpe_train_sharded (include/ecgbyte.h, ecgb_bpe_shard_*), knowing about its neighbours only what the
protocol hands it: the id before its slice, the three ids after it, and the parity of a run of `left` entering it.  The gloo tests
run the REAL exchange loop over it (world size 2 and 3) and compare the outcome with the single-process oracle trainer
(oracle/oracle.py, the restatement of ecg_byte/rust_bpe/src/lib.rs:58-125): if the 8-word summaries, the three-id halo, the parity
fold or the six-vector delta slab lost information, the merges or the ids would differ."""
import numpy as np
import torch

EMPTY = 0xFFFFFFFF


class CpuShard:
    def __init__(self, text: bytes, num_merges: int):
        self.ids = list(text)
        self.V = 256 + int(num_merges)
        self._table = torch.zeros(self.V * self.V, dtype=torch.int64)
        self._slab = torch.zeros(6 * self.V, dtype=torch.int64)
        self.pairs = []
        self.active = True
        self.l = self.r = self.x = EMPTY
        self.pending = None            # ids after the merge in flight (committed by the next pick / finish)

    def new_words(self, n):
        return torch.zeros(n, dtype=torch.int64)

    def table(self):
        return self._table

    def slab(self):
        return self._slab

    # ---- summary of the slice for the merge in flight -----------------------------------------------------------------------
    def _summary(self, out, with_pair):
        a = self.ids
        n = len(a)
        out[0] = n
        for k in range(3):
            out[1 + k] = a[k] if k < n else EMPTY
        out[4] = a[-1] if n else EMPTY
        all_l = tail = 0
        if with_pair and self.active and self.l == self.r and n:
            t = 0
            while t < n and a[n - 1 - t] == self.l:
                t += 1
            all_l, tail = int(t == n), t & 1
        out[5], out[6], out[7] = all_l, tail, 0

    def _halo(self, gathered, rank, world, with_pair):
        g = gathered.view(world, 8).tolist()
        prev, par = EMPTY, 0
        for q in range(rank):
            if g[q][0] == 0:
                continue
            prev = g[q][4]
            par = par ^ (g[q][0] & 1) if g[q][5] else g[q][6]
        lead = par if (with_pair and self.active and self.l == self.r and prev == self.l) else 0
        nxt = []
        for q in range(rank + 1, world):
            for k in range(min(3, g[q][0])):
                if len(nxt) < 3:
                    nxt.append(g[q][1 + k])
        nxt += [EMPTY] * (3 - len(nxt))
        return prev, nxt, lead

    # ---- the steps ---------------------------------------------------------------------------------------------------------------
    def begin(self, summary):
        self._summary(summary, False)

    def count(self, gathered, rank, world):
        _, nxt, _ = self._halo(gathered, rank, world, False)
        a = self.ids
        for k in range(len(a)):
            b = a[k + 1] if k + 1 < len(a) else nxt[0]
            if b != EMPTY:
                self._table[a[k] * self.V + b] += 1           # the LEFT rank of a pair counts it

    def pick(self, i, summary):
        if self.pending is not None:
            self.ids, self.pending = self.pending, None
        if self.active:
            t = self._table.view(self.V, self.V)[: 256 + i]
            top = int(t.max())
            if top > 0:
                idx = int(torch.nonzero(t.reshape(-1) == top)[0])   # first = smallest (left, right)
                self.l, self.r, self.x = idx // self.V, idx % self.V, 256 + i
                self.pairs.append((self.l, self.r))
            else:
                self.active = False
        self._summary(summary, True)

    def _slab_index(self, a, b):
        V = self.V
        if a == self.l: return b
        if a == self.r: return V + b
        if a == self.x: return 2 * V + b
        if b == self.l: return 3 * V + a
        if b == self.r: return 4 * V + a
        assert b == self.x, (a, b, self.l, self.r, self.x)
        return 5 * V + a

    def merge(self, i, gathered, rank, world):
        if not self.active:
            return
        prev, nxt, lead = self._halo(gathered, rank, world, True)
        l, r, X = self.l, self.r, self.x
        a = self.ids
        n = len(a)
        ext = [prev] + a + nxt                              # ext[p + 1] = a[p]; ext[0] = the id before, ext[n + 1 ..] the three after
        # site[p]: a merge starts at ext position p (0 .. n + 2)
        site = [False] * (n + 4)
        if l != r:
            for p in range(n + 3):
                site[p] = ext[p] == l and ext[p + 1] == r
        else:
            q = lead                                          # run offset parity of a[0]
            site[0] = prev == l and lead == 1 and n > 0 and a[0] == l
            for p in range(1, n + 4):
                if ext[p] == l:
                    if q == 0 and p + 1 < n + 4 and ext[p + 1] == l:
                        site[p] = True
                    q ^= 1
                else:
                    q = 0
        second = [p >= 1 and site[p - 1] for p in range(n + 4)]
        out = []
        for p in range(1, n + 1):                             # local elements
            if ext[p + 1] != EMPTY and (site[p - 1] or site[p] or site[p + 1]):
                self._slab[self._slab_index(ext[p], ext[p + 1])] -= 1
            if second[p]:
                continue
            val = X if site[p] else ext[p]
            out.append(val)
            kn = p + 2 if site[p] else p + 1                  # next kept element
            an = ext[kn] if kn < n + 4 else EMPTY
            if an != EMPTY:
                nval = X if site[kn] else an
                if site[p] or site[kn]:
                    self._slab[self._slab_index(val, nval)] += 1
        self.pending = out

    def apply(self):
        if self.active:
            V = self.V
            fixed = [self.l, self.r, self.x]
            for j in torch.nonzero(self._slab).view(-1).tolist():
                part, o = divmod(j, V)
                f = fixed[part % 3]
                a, b = (f, o) if part < 3 else (o, f)
                self._table[a * V + b] += self._slab[j]
        self._slab.zero_()

    def finish(self):
        if self.pending is not None:
            self.ids, self.pending = self.pending, None
        return list(self.ids), list(self.pairs)
