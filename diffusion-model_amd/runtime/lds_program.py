"""Host-side builder of the programs the LDS-resident UNet kernel runs (csrc/unet_lds.hip, ddpm_unet_lds_run).

A program is a table of operations on tiles of one workgroup's LDS arena: the builder hands out tiles, appends operations, lays the arena out once
every tile's lifetime is known, and returns the table as a ctypes array.  A tile holds the pixels of the workgroup's
images (image-major rows) with bf16 channels; the row stride is the channel bytes plus 16, which spreads the 32 rows a
matrix-core fragment load touches over the LDS banks.

The reference expresses the same work as nn.Module calls, one torch kernel per layer (unets.py:127-131,230-237;
constituent_blocks.py:135-147,227-243; conv_block.py:132-136 there)."""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

import torch
from torch import Tensor

from . import hip

K33, K33S2, K11, KT22 = hip.CONV_3X3_S1, hip.CONV_3X3_S2, hip.CONV_1X1, hip.CONVT_2X2_S2


class LdsOutOfSpace(RuntimeError):
    pass


class Tile:
    """rows = ipw * w * w pixels of `ch` bf16 channels; `off` (byte offset in the arena) is assigned when the program is
    laid out.  Live from operation `birth` up to (not including) operation `death`."""
    __slots__ = ("off", "stride", "w", "ch", "nbytes", "birth", "death")

    def __init__(self, stride: int, w: int, ch: int, nbytes: int, birth: int):
        self.off, self.stride, self.w, self.ch, self.nbytes, self.birth, self.death = -1, stride, w, ch, nbytes, birth, None


class LdsProgram:
    def __init__(self, ipw: int, act: int):
        self.ipw, self.act = int(ipw), int(act)
        lib = hip.lib()
        self.arena0, self.red_bytes, self.max_bytes = lib.ddpm_unet_lds_arena0(), lib.ddpm_unet_lds_reduce_bytes(), lib.ddpm_unet_lds_max_bytes()
        self.tiles: List[Tile] = []
        self.ops: List[hip.LdsOp] = []
        self.refs = []                        # (operation, offset field, tile): patched when the arena is laid out
        self.keep: List[Tensor] = []          # tensors the table points at
        self.peak = self.arena0
        self._laid_out = False

    # ---- arena -----------------------------------------------------------------------------------------------------
    def tile(self, w: int, ch: int) -> Tile:
        stride = ch * 2 + 16
        t = Tile(stride, w, ch, (self.ipw * w * w * stride + 15) // 16 * 16, len(self.ops))
        self.tiles.append(t)
        self._laid_out = False
        return t

    def _scratch(self, nbytes: int) -> Tile:
        """bytes that live for the next operation only"""
        t = Tile(0, 0, 0, (nbytes + 15) // 16 * 16, len(self.ops))
        t.death = len(self.ops) + 1
        self.tiles.append(t)
        return t

    def free(self, t: Tile) -> None:
        t.death = len(self.ops)

    def layout(self) -> int:
        """Assign arena offsets to all tiles, knowing every lifetime: largest tile first, each at the lowest offset where it
        overlaps no already placed tile that is live at the same time (the greedy-by-size rule of offline memory planners:
        the few big tiles of a program, a stage's input image or a transposed convolution's output, get the arena's ends
        and the small ones fill in around them).  Returns the bytes of LDS the program needs."""
        if self._laid_out:
            return self.peak
        end = len(self.ops) + 1                # (tiles never freed live to the end; a tile made after the last operation still counts)
        placed: List[Tile] = []
        self.peak = self.arena0
        for t in sorted(self.tiles, key=lambda t: (-t.nbytes, t.birth)):
            d = end if t.death is None else t.death
            busy = sorted((q.off, q.off + q.nbytes) for q in placed
                          if q.birth < d and t.birth < (end if q.death is None else q.death))
            at = self.arena0
            for lo, hi in busy:
                if lo - at >= t.nbytes:
                    break
                at = max(at, hi)
            if at + t.nbytes > self.max_bytes:
                raise LdsOutOfSpace(f"the program does not fit {self.max_bytes} bytes of LDS ({t.nbytes} more bytes at operation {t.birth})")
            t.off = at
            placed.append(t)
            self.peak = max(self.peak, at + t.nbytes)
        for op, field, t in self.refs:
            setattr(op, field, t.off)
        self._laid_out = True
        return self.peak

    # ---- operations ------------------------------------------------------------------------------------------------
    def _op(self, **kw) -> hip.LdsOp:
        op = hip.LdsOp()
        ptrs = kw.pop("p", ())
        for field in ("src", "dst", "aux"):
            t = kw.pop(field, None)
            if t is not None:
                self.refs.append((op, field + "_off", t))
                if t.stride:
                    setattr(op, field + "_stride", t.stride)
        self._laid_out = False
        for k, v in kw.items():
            setattr(op, k, v)
        for i, t in enumerate(ptrs):
            if t is not None:
                op.p[i] = t.data_ptr()
                self.keep.append(t)
        self.ops.append(op)
        return op

    @staticmethod
    def _nhwc(g: Tensor):
        assert g.dtype == torch.bfloat16 and g.dim() == 4 and g.is_contiguous(), "LDS program tensors are contiguous NHWC bf16"
        return g.shape[3]

    def load(self, t: Tile, g: Tensor, col0: int = 0, ch: Optional[int] = None, gcol0: int = 0) -> None:
        gs = self._nhwc(g)
        self._op(type=hip.LOP_LOAD, dst=t, w_out=t.w, cout=ch or gs, col0=col0, gstride=gs, gcol0=gcol0, p=(g,))

    def store(self, t: Tile, g: Tensor, col0: int = 0, ch: Optional[int] = None, gcol0: int = 0) -> None:
        gs = self._nhwc(g)
        self._op(type=hip.LOP_STORE, src=t, w_out=t.w, cout=ch or gs, col0=col0, gstride=gs, gcol0=gcol0, p=(g,))

    def addt(self, t: Tile, temb: Tensor, ch: int, col0: int = 0) -> None:
        assert temb.dtype == torch.float32 and temb.is_contiguous() and temb.shape[-1] == ch
        self._op(type=hip.LOP_ADDT, dst=t, w_out=t.w, cout=ch, col0=col0, p=(temb,))

    def mask(self, src: Tile, dst: Tile, bits: Optional[Tensor], ch: int, scale: float = 1.0) -> None:
        """dst columns [0, ch) = src columns [0, ch) * keep bits * scale; bits = None: a plain copy"""
        self._op(type=hip.LOP_MASK, src=src, dst=dst, w_out=src.w, cout=ch,
                 mscale=scale, p=(bits,))

    def conv(self, kind: int, src: Tile, dst: Tile, w: Tensor, bias: Optional[Tensor], cin: int, cout: int, col0: int = 0) -> None:
        """src columns [0, cin) -> dst columns [col0, col0 + cout); w = the forward weight copy bf16 [taps][rows][cin]"""
        assert w.dtype == torch.bfloat16 and w.dim() == 3 and w.shape[2] == cin and w.shape[1] >= cout, (tuple(w.shape), cin, cout)
        rm = self.ipw * (src.w * src.w if kind == KT22 else dst.w * dst.w)
        tiles = (4 if kind == KT22 else 1) * ((rm + 31) // 32) * (cout // 32)
        scratch = self._scratch(self.red_bytes) if tiles < 8 else None
        self._op(type=hip.LOP_CONV, kind=kind, src=src, dst=dst, aux=scratch, w_in=src.w, w_out=dst.w, cin=cin, cout=cout, col0=col0,
                 gstride=w.shape[1], p=(w, bias))

    def gn(self, t: Tile, gamma: Tensor, beta: Tensor, groups: int, act: int, ch: int, col0: int = 0, res: Optional[Tile] = None,
           mask: Optional[Tensor] = None, mscale: float = 1.0, save_raw: Optional[Tensor] = None, save_y: Optional[Tensor] = None,
           save_stats: Optional[Tensor] = None, eps: float = 1e-5) -> None:
        assert act in (0, self.act), "a launch is built for one activation"
        self._op(type=hip.LOP_GN, dst=t, aux=res, w_out=t.w, cout=ch, col0=col0, groups=groups, act=act, eps=eps, mscale=mscale,
                 p=(gamma, beta, mask, save_raw, save_y, save_stats))

    def dw(self, t: Tile, w9c: Tensor, bias: Tensor, ch: int) -> None:
        assert w9c.dtype == torch.float32 and tuple(w9c.shape) == (9, ch)
        self._op(type=hip.LOP_DW, dst=t, w_out=t.w, cout=ch, p=(w9c, bias))

    # ---- table -----------------------------------------------------------------------------------------------------
    def table(self):
        """(ctypes array of the operations closed by LOP_END, its size in bytes)"""
        self.layout()
        n = len(self.ops) + 1
        arr = (hip.LdsOp * n)(*self.ops, hip.LdsOp())
        return arr, C.sizeof(arr)

    def check(self) -> int:
        """validate on the host; returns the LDS bytes the program needs"""
        arr, _ = self.table()
        need = hip.lib().ddpm_unet_lds_check(arr, len(arr), self.ipw)
        if need < 0:
            raise RuntimeError("unet_lds program rejected: " + hip.lib().ddpm_last_error().decode())
        return int(need)

    def run_blocking(self, n: int) -> None:
        """Upload the table with an ordinary (synchronising) copy and launch: tests and one-off calls.  The engine stages its
        tables through pinned memory instead, so that a launch can be captured in a HIP graph."""
        arr, nbytes = self.table()
        dev = self.keep[0].device
        blob = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        hip.call("ddpm_unet_lds_run", blob.data_ptr(), arr, len(arr), n, self.ipw, self.act, hip.stream())
        torch.cuda.current_stream().synchronize()
