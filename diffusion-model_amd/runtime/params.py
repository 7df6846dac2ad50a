"""Device-side bookkeeping of the denoiser's parameters: kernel-layout weight copies, gradient buffers and the remap
tables that convert between those and the torch layouts the nn.Parameters keep (one kernel launch per direction).

Layouts.  Forward weight copy  : bf16 [slices][CoutW][CinP]   (CoutW = Cout rounded up to 32, CinP = Cin to 16)
          Data-gradient copy   : bf16 [slices][CinW][CoutP]   (rows = input channels; 3x3 stride-1 taps flipped)
          Depthwise weights    : fp32 [9][C] and its tap-flipped twin
          Gradients (kernel)   : fp32 [slices][Cout][CinP] / [4][Cout][Cin] / [9][C] / [C]
          Gradients (torch)    : one flat fp32 buffer, each parameter's .grad is a view of it in the parameter's layout
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Sequence, Tuple

import torch
from torch import Tensor

from . import hip

K33, K33S2, K11, KT22 = hip.CONV_3X3_S1, hip.CONV_3X3_S2, hip.CONV_1X1, hip.CONVT_2X2_S2


# Bumped by optimizers that update parameters with a raw kernel (which torch's per-tensor version counters cannot see),
# so that cached kernel-layout weight copies are refreshed on the next forward.
_WEIGHT_EPOCH = [0]


def bump_weight_epoch() -> None:
    _WEIGHT_EPOCH[0] += 1


def _ru(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def _entry(a: int, b: int, S, R, K, Rv, Kv, a_r, a_k, off_s: Sequence[int], a_bf16: bool) -> hip.RemapEntry:
    e = hip.RemapEntry(a=a, b=b, S=S, R=R, K=K, Rv=Rv, Kv=Kv, a_r=a_r, a_k=a_k, a_bf16=int(a_bf16), pad=0)
    for i in range(9):
        e.off_s[i] = off_s[i] if i < len(off_s) else 0
    return e


def _to_device(entries: List[hip.RemapEntry], device) -> Tuple[Tensor, Tensor, int]:
    """(entry table, chunk table, number of chunks) on the device"""
    for e in entries:
        if e.S * e.R * e.K >= 1 << 31 or e.Rv * e.a_r + e.Kv * e.a_k + max(e.off_s) >= 1 << 31:
            raise ValueError("a parameter tensor is too large for the remap kernel's 32-bit indices")
    arr = (hip.RemapEntry * len(entries))(*entries)
    t = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    ce = hip.lib().ddpm_remap_chunk_elems()
    chunks = []
    for i, e in enumerate(entries):
        n = e.S * e.R * e.K
        chunks.extend((i, c) for c in range((n + ce - 1) // ce))
    ct = torch.tensor(chunks, dtype=torch.int32).to(device)
    return t, ct, len(chunks)


def lookup(params: Dict[str, Tensor], aliases: Dict[str, Tuple[str, int, int]], key: str) -> Tensor:
    """params[key], or - for a logical name that stands for a row range of a packed parameter (the q / k,v blocks of
    nn.MultiheadAttention's in_proj_weight, or its q_proj_weight-style names) - that row range as a view."""
    if key in params:
        return params[key]
    real, r0, rows = aliases[key]
    return params[real][r0:r0 + rows]


def conv_dims(kind: int, w: Tensor) -> Tuple[int, int, int]:
    """(Cout, Cin, taps) of a dense convolution / linear weight in torch layout."""
    if kind == KT22:
        return w.shape[1], w.shape[0], 4
    kk = 1 if w.dim() == 2 else w.shape[2] * w.shape[3]
    return w.shape[0], w.shape[1], kk


class WeightCache:
    """Kernel-layout copies of all convolution weights, refreshed by ONE remap launch."""

    def __init__(self, convs: Dict[str, int], dwconvs: List[str], aliases: Dict[str, Tuple[str, int, int]] = None):
        self.convs, self.dwconvs, self.aliases = convs, dwconvs, aliases or {}
        self.fwd: Dict[str, Tensor] = {}
        self.dgrad: Dict[str, Tensor] = {}
        self.dw: Dict[str, Tensor] = {}
        self.dw_flip: Dict[str, Tensor] = {}
        self._ptrs: Tuple[int, ...] = ()
        self._vers: Tuple[int, ...] = ()
        self._table = self._chunks = None
        self._n = 0

    def _build(self, params: Dict[str, Tensor], device) -> None:
        entries: List[hip.RemapEntry] = []
        owner: List[str] = []                         # layer name of every entry (refresh_where)
        for name, kind in self.convs.items():
            owner += [name, name]
            w = lookup(params, self.aliases, name + ".weight")
            co, ci, kk = conv_dims(kind, w)
            cow, cip, ciw, cop = _ru(co, 32), _ru(ci, 16), _ru(ci, 32), _ru(co, 16)
            if name not in self.fwd:
                self.fwd[name] = torch.zeros(kk, cow, cip, device=device, dtype=torch.bfloat16)
                self.dgrad[name] = torch.zeros(kk, ciw, cop, device=device, dtype=torch.bfloat16)
            ident = list(range(kk))
            if kind == KT22:
                entries.append(_entry(self.fwd[name].data_ptr(), w.data_ptr(), kk, cow, cip, co, ci, 4, co * 4, ident, True))
                entries.append(_entry(self.dgrad[name].data_ptr(), w.data_ptr(), kk, ciw, cop, ci, co, co * 4, 4, ident, True))
            else:
                flip = [kk - 1 - s for s in range(kk)] if kind == K33 else ident
                entries.append(_entry(self.fwd[name].data_ptr(), w.data_ptr(), kk, cow, cip, co, ci, ci * kk, kk, ident, True))
                entries.append(_entry(self.dgrad[name].data_ptr(), w.data_ptr(), kk, ciw, cop, ci, co, kk, ci * kk, flip, True))
        for name in self.dwconvs:
            owner += [name, name]
            w = params[name + ".weight"]
            c = w.shape[0]
            if name not in self.dw:
                self.dw[name] = torch.zeros(9, c, device=device, dtype=torch.float32)
                self.dw_flip[name] = torch.zeros(9, c, device=device, dtype=torch.float32)
            entries.append(_entry(self.dw[name].data_ptr(), w.data_ptr(), 9, 1, c, 1, c, 0, 9, list(range(9)), False))
            entries.append(_entry(self.dw_flip[name].data_ptr(), w.data_ptr(), 9, 1, c, 1, c, 0, 9, [8 - s for s in range(9)], False))
        self._table, self._chunks, self._n = _to_device(entries, device)
        self._entries, self._owner, self._subsets = entries, owner, {}

    def _state(self, params: Dict[str, Tensor]):
        names = [n + ".weight" for n in self.convs] + [n + ".weight" for n in self.dwconvs]
        ws = [lookup(params, self.aliases, n) for n in names]
        return tuple(w.data_ptr() for w in ws), (_WEIGHT_EPOCH[0],) + tuple(w._version for w in ws)

    def refresh(self, params: Dict[str, Tensor], device, force: bool = False) -> None:
        ptrs, vers = self._state(params)
        if ptrs != self._ptrs:
            self._build(params, device)
            self._ptrs, self._vers = ptrs, ()
        if force or vers != self._vers:
            hip.call("ddpm_remap", hip.ptr(self._table), hip.ptr(self._chunks), self._n, 1, hip.stream())
            self._vers = vers

    def refresh_where(self, tag: str, keep, params: Dict[str, Tensor], device, done: bool) -> None:
        """refresh() for the layers whose name passes keep(name) (cached under `tag`), for a driver that updates the parameters
        in pieces: one launch over that subset.  done=True on the piece that completes the set: the copies then count as
        current for the parameters' present state."""
        ptrs, vers = self._state(params)
        if ptrs != self._ptrs:
            raise RuntimeError("WeightCache.refresh_where: the parameters moved since the copies were laid out; call refresh() first")
        if tag not in self._subsets:
            sub = [e for e, n in zip(self._entries, self._owner) if keep(n)]
            self._subsets[tag] = _to_device(sub, device) if sub else None
        t = self._subsets[tag]
        if t is not None:
            hip.call("ddpm_remap", hip.ptr(t[0]), hip.ptr(t[1]), t[2], 1, hip.stream())
        if done:
            self._vers = vers


class GradStore:
    """fp32 gradient accumulators in kernel layout (one flat zero-initialised buffer) and their torch-layout image."""

    def __init__(self, convs: Dict[str, int], dwconvs: List[str], params: Dict[str, Tensor], device,
                 aliases: Dict[str, Tuple[str, int, int]] = None):
        self.aliases = aliases or {}
        self.layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        self.tlayout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        off = toff = 0
        recipe = []
        for name, p in params.items():
            base = name.rsplit(".", 1)[0]
            n_t = p.numel()
            if name.endswith(".weight") and base in convs:
                kind = convs[base]
                co, ci, kk = conv_dims(kind, p)
                if kind == KT22:
                    shp = (4, co, ci)
                    recipe.append((off, toff, 4, co, ci, co, ci, 4, co * 4, list(range(4))))
                else:
                    cip = _ru(ci, 16)
                    shp = (kk, co, cip)
                    recipe.append((off, toff, kk, co, cip, co, ci, ci * kk, kk, list(range(kk))))
            elif name.endswith(".weight") and base in dwconvs:
                c = p.shape[0]
                shp = (9, c)
                recipe.append((off, toff, 9, 1, c, 1, c, 0, 9, list(range(9))))
            else:
                shp = (n_t,)
                recipe.append((off, toff, 1, 1, n_t, 1, n_t, 0, 1, [0]))
            n_k = 1
            for s in shp:
                n_k *= s
            self.layout[name] = (off, shp)
            self.tlayout[name] = (toff, tuple(p.shape))
            off += _ru(n_k, 4)
            toff += _ru(n_t, 4)
        self.flat = torch.zeros(off, device=device, dtype=torch.float32)
        self.flat_t = torch.zeros(toff, device=device, dtype=torch.float32)
        self.views = {k: self.flat[o:o + _numel(s)].view(s) for k, (o, s) in self.layout.items()}
        fa, fb = self.flat.data_ptr(), self.flat_t.data_ptr()
        entries = [_entry(fa + 4 * o, fb + 4 * t, S, R, K, Rv, Kv, a_r, a_k, offs, False) for (o, t, S, R, K, Rv, Kv, a_r, a_k, offs) in recipe]
        self._entries, self._names, self._device = entries, list(params.keys()), device
        self._table, self._chunks, self._n = _to_device(entries, device)
        self._phase_tables: Dict = {}

    def zero(self):
        self.flat.zero_()

    def _alias(self, name: str) -> Tuple[int, int, int]:
        """(offset, rows, row length) of a logical weight that is a row range of a parameter kept in its torch layout
        (out x in, which is also the kernel layout of a pointwise weight gradient when `in` is a multiple of 16)"""
        real, r0, rows = self.aliases[name]
        off, shp = self.layout[real]
        k = _numel(self.tlayout[real][1][1:])
        if len(shp) != 1 or k % 16:
            raise ValueError(f"{real}: a row range of this parameter cannot be addressed in the gradient buffer")
        return off + r0 * k, rows, k

    def offset(self, name: str) -> int:
        if name not in self.layout:
            return self._alias(name)[0]
        return self.layout[name][0]

    def __getitem__(self, name: str) -> Tensor:
        if name not in self.views:
            off, rows, k = self._alias(name)
            self.views[name] = self.flat[off:off + rows * k].view(1, rows, k)
        return self.views[name]

    def unpack_phase(self, prefix: str, phase: int) -> Tuple[int, int]:
        """Kernel layout -> torch layout (into the persistent staging buffer flat_t, no copy) for one half of the parameters:
        phase 0 = those whose name starts with `prefix`, phase 1 = the others.  Returns the element range [lo, hi) of flat_t
        that the half occupies (it must be contiguous: parameters are laid out in registration order)."""
        key = (prefix, phase)
        if key not in self._phase_tables:
            idx = [i for i, n in enumerate(self._names) if n.startswith(prefix) == (phase == 0)]
            if not idx or idx != list(range(idx[0], idx[-1] + 1)):
                raise ValueError(f"parameters starting with {prefix!r} do not form one contiguous, non-empty run")
            tab, chunks, n = _to_device([self._entries[i] for i in idx], self._device)
            lo = self.tlayout[self._names[idx[0]]][0]
            last = self.tlayout[self._names[idx[-1]]]
            self._phase_tables[key] = (tab, chunks, n, lo, last[0] + _ru(_numel(last[1]), 4))
        tab, chunks, n, lo, hi = self._phase_tables[key]
        hip.call("ddpm_remap", hip.ptr(tab), hip.ptr(chunks), n, 0, hip.stream())
        return lo, hi

    def unpack_where(self, tag: str, keep) -> List[Tuple[int, int]]:
        """unpack_phase for an arbitrary subset of the parameters (keep(name) -> bool, cached under `tag`): one launch;
        returns the element ranges [lo, hi) of flat_t the subset occupies, consecutive parameters merged into one range."""
        if tag not in self._phase_tables:
            idx = [i for i, n in enumerate(self._names) if keep(n)]
            if not idx:
                self._phase_tables[tag] = None
            else:
                tab, chunks, n = _to_device([self._entries[i] for i in idx], self._device)
                ranges, start, prev = [], idx[0], idx[0]
                for i in idx[1:] + [None]:
                    if i is None or i != prev + 1:
                        last = self.tlayout[self._names[prev]]
                        ranges.append((self.tlayout[self._names[start]][0], last[0] + _ru(_numel(last[1]), 4)))
                        start = i
                    prev = i
                self._phase_tables[tag] = (tab, chunks, n, ranges)
        t = self._phase_tables[tag]
        if t is None:
            return []
        hip.call("ddpm_remap", hip.ptr(t[0]), hip.ptr(t[1]), t[2], 0, hip.stream())
        return list(t[3])

    def torch_views(self) -> Dict[str, Tensor]:
        """{name: view of the persistent torch-layout staging buffer}: what unpack_phase fills (callers that bind these as
        .grad must consume them before the next backward pass overwrites the buffer)"""
        return {k: self.flat_t[o:o + _numel(shp)].view(shp) for k, (o, shp) in self.tlayout.items()}

    def unpack_staging(self) -> Tensor:
        """kernel layout -> torch layout for all parameters in one launch, into the persistent staging buffer (returned):
        the next backward pass overwrites it."""
        hip.call("ddpm_remap", hip.ptr(self._table), hip.ptr(self._chunks), self._n, 0, hip.stream())
        return self.flat_t

    def unpack(self) -> Dict[str, Tensor]:
        """kernel layout -> torch layout for all parameters in one launch; returns {name: gradient}.  The gradients are
        views of ONE freshly allocated flat buffer (a copy of the staging buffer), so a caller - autograd's accumulation
        step in particular - may keep them: the next backward pass will not overwrite them."""
        hip.call("ddpm_remap", hip.ptr(self._table), hip.ptr(self._chunks), self._n, 0, hip.stream())
        out = self.flat_t.clone()
        return {k: out[o:o + _numel(shp)].view(shp) for k, (o, shp) in self.tlayout.items()}


def flat_grad_view(params: Sequence[Tensor], offs: Sequence[int], total: int):
    """If the gradients of `params` all live in ONE fp32 allocation, parameter i at element offset offs[i] of a
    `total`-element buffer (the layout GradStore.unpack produces), return that buffer as a flat tensor (zero-copy);
    otherwise None.  autograd detaches the views it is handed, so the test is on storage identity and offsets."""
    g0 = params[0].grad
    if g0 is None or g0.dtype != torch.float32:
        return None
    st = g0.untyped_storage()
    if st.nbytes() != total * 4:
        return None
    key = st.data_ptr()
    for p, o in zip(params, offs):
        g = p.grad
        if g is None or g.dtype != torch.float32 or g.untyped_storage().data_ptr() != key or g.storage_offset() != o or not g.is_contiguous():
            return None
    return torch.empty(0, device=g0.device, dtype=torch.float32).set_(st, 0, (total,))


def _numel(shape) -> int:
    n = 1
    for s in shape:
        n *= s
    return n
