"""Batch form of ``HyperbolicTokenizer.tokenize`` / ``encode`` on the GPU (C-ABI ``hm_tokenize_batch``).

Reference semantics (tokenizer/hyperbolic_merge.py:414-459): start from ``list(text)``, apply the rules
``{(old1, old2): new}`` of ``merge_history`` in repeated left-to-right passes (a hit rewrites position i, drops
position i+1 and stays at i) until a pass changes nothing; ``encode`` maps every token through ``token2idx`` with
``<unk>`` (or 3) for strings the vocabulary does not hold.  The reference drives this one line at a time
(scripts/benchmark_efficiency.py:58-94); here all lines of a batch go through one kernel launch, one lane per line.

Host side of the boundary (this file): strings <-> 32-bit symbols.  Every distinct string a rule mentions and every
single-character vocabulary entry gets a symbol >= 0; any other character c travels as ``-(2 + ord(c))``, can never
match a rule and comes back as itself.  There is no CPU path: without libhypmerge.so the constructor raises.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from .. import _lib

_N_CODEPOINTS = 0x110000


class BatchEncoder:
    """Compiled rule set of one tokenizer state, bound to one GPU."""

    def __init__(self, rules: Dict[Tuple[str, str], str], token2idx: Dict[str, int], device: torch.device):
        self._L = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.HypMergeUnavailable("BatchEncoder needs a GPU device (there is no CPU path)")
        self.unk = int(token2idx.get("<unk>", 3))
        sym: Dict[str, int] = {}
        for (a, b), ab in rules.items():
            for s in (a, b, ab):
                if s not in sym:
                    sym[s] = len(sym)
        for s in token2idx:
            if len(s) == 1 and s not in sym:
                sym[s] = len(sym)
        self.strings: List[str] = [""] * len(sym)
        for s, k in sym.items():
            self.strings[k] = s
        n_sym = len(sym)
        if n_sym >= (1 << 21) - 1:
            raise ValueError(f"{n_sym} distinct strings: the rule table packs symbols into 21 bits")
        # character -> symbol (dense over the code space: one gather per batch)
        lut = -(2 + np.arange(_N_CODEPOINTS, dtype=np.int64))
        for s, k in sym.items():
            if len(s) == 1:
                lut[ord(s)] = k
        self._lut = lut.astype(np.int32)
        # symbol -> vocabulary index (encode): unk where the string is not a vocabulary entry
        self._sym2vocab = np.array([token2idx.get(s, self.unk) for s in self.strings], dtype=np.int64)
        self._strings_arr = np.array(self.strings, dtype=object)
        # rule table
        n_rules = len(rules)
        left = np.fromiter((sym[a] for (a, _b) in rules), dtype=np.int32, count=n_rules)
        right = np.fromiter((sym[b] for (_a, b) in rules), dtype=np.int32, count=n_rules)
        merged = np.fromiter((sym[ab] for ab in rules.values()), dtype=np.int32, count=n_rules)
        cap = int(self._L.hm_tokenize_table_capacity(n_rules))
        table = np.empty(cap, dtype=np.uint64)
        _lib.check(self._L.hm_tokenize_build_table(left.ctypes.data, right.ctypes.data, merged.ctypes.data, n_rules,
                                                   table.ctypes.data, cap))
        self.capacity = cap
        self.n_rules = n_rules
        self._table = torch.from_numpy(table.view(np.int64)).to(self.device)      # torch allocations are 256-byte aligned

    # ------------------------------------------------------------------------------------------
    def symbols(self, texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """(symbols of all lines concatenated, offsets[n + 1]) -- list(text) of the reference, as numbers."""
        lens = np.fromiter((len(t) for t in texts), dtype=np.int64, count=len(texts))
        offsets = np.zeros(len(texts) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        cps = np.frombuffer("".join(texts).encode("utf-32-le", "surrogatepass"), dtype=np.uint32)
        if cps.shape[0] != offsets[-1]:
            raise ValueError("text length mismatch after UTF-32 encoding")
        return self._lut[cps], offsets

    def run(self, sym: torch.Tensor, offsets: torch.Tensor, order: Optional[torch.Tensor] = None, want_passes: bool = False):
        """Device arrays in, device arrays out: (out, out_len, passes | None).  Asynchronous on the current stream."""
        n = offsets.numel() - 1
        out = torch.empty_like(sym)
        out_len = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device)
        passes = torch.zeros(max(n, 1), dtype=torch.int32, device=self.device) if want_passes else None
        stream = torch.cuda.current_stream(self.device).cuda_stream
        _lib.check(self._L.hm_tokenize_batch(
            sym.data_ptr() if sym.numel() else None, offsets.data_ptr(), order.data_ptr() if order is not None else None, n,
            self._table.data_ptr(), self.capacity,
            out.data_ptr() if out.numel() else None, out_len.data_ptr(), passes.data_ptr() if passes is not None else None,
            C.c_void_p(stream)))
        return out, out_len[:n], (passes[:n] if passes is not None else None)

    def _tokens(self, texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """(flat symbols after the merges, lengths per line) on the host."""
        n = len(texts)
        if n == 0:
            return np.zeros(0, dtype=np.int32), np.zeros(0, dtype=np.int64)
        sym_h, off_h = self.symbols(texts)
        lens_h = np.diff(off_h)
        if lens_h.max() >= 2 ** 31:
            raise ValueError("a line of 2^31 or more characters")
        with torch.cuda.device(self.device):
            sym = torch.from_numpy(sym_h).to(self.device)
            off = torch.from_numpy(off_h).to(self.device)
            lens = off[1:] - off[:-1]
            order = torch.argsort(lens, descending=True, stable=True)
            out, out_len, _ = self.run(sym, off, order)
            if sym.numel():
                pos = torch.arange(sym.numel(), device=self.device) - torch.repeat_interleave(off[:-1], lens)
                keep = pos < torch.repeat_interleave(out_len.to(torch.int64), lens)
                flat = out[keep].cpu().numpy()
            else:
                flat = np.zeros(0, dtype=np.int32)
            out_lens = out_len.cpu().numpy().astype(np.int64)
        assert flat.shape[0] == int(out_lens.sum()) and lens_h.shape[0] == n
        return flat, out_lens

    def tokenize_batch(self, texts: Sequence[str]) -> List[List[str]]:
        flat, lens = self._tokens(texts)
        strs = np.empty(flat.shape[0], dtype=object)
        known = flat >= 0
        strs[known] = self._strings_arr[flat[known]]
        if not known.all():
            strs[~known] = [chr(-int(s) - 2) for s in flat[~known]]
        ends = np.cumsum(lens)
        lst = strs.tolist()
        return [lst[int(e - k):int(e)] for e, k in zip(ends, lens)]

    def encode_arrays(self, texts: Sequence[str]) -> Tuple[np.ndarray, np.ndarray]:
        """``encode`` of every line as (ids of all lines concatenated, offsets[n + 1]) -- no Python lists."""
        flat, lens = self._tokens(texts)
        ids = np.full(flat.shape[0], self.unk, dtype=np.int64)
        known = flat >= 0
        ids[known] = self._sym2vocab[flat[known]]
        offsets = np.zeros(len(texts) + 1, dtype=np.int64)
        np.cumsum(lens, out=offsets[1:])
        return ids, offsets

    def encode_batch(self, texts: Sequence[str]) -> List[List[int]]:
        ids, offsets = self.encode_arrays(texts)
        lst = ids.tolist()
        return [lst[int(b):int(e)] for b, e in zip(offsets[:-1], offsets[1:])]
