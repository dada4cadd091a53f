#!/usr/bin/env python3
"""Writes cuda-matrix-inversion_amd/csrc/gather_tree.inc: the B-operand gather of ONE pivot row out of a wave's accumulator tiles
(run-time register index) as inline asm with a BINARY branch tree.

What it replaces: tilep_impl.hpp's gather_zero_tile_row* -- one asm block per tile row that branches over itself unless the pivot
lives there. Stamped on the GPU (r03, s_memtime in tilepb_impl.hpp): a skipped block costs ~75 cycles (three scalar instructions
and a TAKEN branch), i.e. ~600 cycles per pivot row at 8 tile rows, more than the search that found the pivot. Here one asm block
covers up to 4 tile rows (one tile column wide) or 3 (two tile columns wide) -- the 30-operand limit of an asm statement -- and
finds the (tile row, register) slot with a binary tree of s_bitcmp1 / s_cbranch over the bits of the slot index: about four taken
branches per pivot row at 8 tile rows instead of ten.

pos = 4 * tile row + register (wave-uniform), mask = the 16 lanes of the lane group that holds the row, addr = LDS byte address of
this lane's element of the row being gathered; the row's registers are zeroed afterwards (C operand of the pivot rows)."""
import os

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "cuda-matrix-inversion_amd", "csrc", "gather_tree.inc")


def chunk(bits, t, r, cols, col_stride):
    w = "ds_write_b64" if bits == 64 else "ds_write_b32"
    z = "v_mov_b64_e32" if bits == 64 else "v_mov_b32_e32"
    s = ""
    for j in range(cols):
        off = f" offset:{j * col_stride}" if j else ""
        s += f'"{w} %[addr], %[a{t}{j}{r}]{off}\\n\\t"\n'
    for j in range(cols):
        s += f'"{z} %[a{t}{j}{r}], 0\\n\\t"\n'
    return s


def reg_tree(bits, t, cols, col_stride, last):
    """register r = idx & 3 of tile t; labels 1t0.. are local to the asm statement (numeric labels)"""
    L = lambda k: f"{1 + t}{k}"
    s = f'"s_bitcmp1_b32 %[idx], 1\\n\\t"\n"s_cbranch_scc1 {L(2)}f\\n\\t"\n'
    s += f'"s_bitcmp1_b32 %[idx], 0\\n\\t"\n"s_cbranch_scc1 {L(1)}f\\n\\t"\n'
    s += chunk(bits, t, 0, cols, col_stride) + '"s_branch 8f\\n"\n'
    s += f'"{L(1)}:\\n\\t"\n' + chunk(bits, t, 1, cols, col_stride) + '"s_branch 8f\\n"\n'
    s += f'"{L(2)}:\\n\\t"\n"s_bitcmp1_b32 %[idx], 0\\n\\t"\n"s_cbranch_scc1 {L(3)}f\\n\\t"\n'
    s += chunk(bits, t, 2, cols, col_stride) + '"s_branch 8f\\n"\n'
    s += f'"{L(3)}:\\n\\t"\n' + chunk(bits, t, 3, cols, col_stride)
    if not last:
        s += '"s_branch 8f\\n"\n'
    else:
        s += '"\\n"\n'
    return s


def tile_tree(bits, tiles, cols, col_stride):
    """tile t = idx >> 2 (bits 2, 3 of idx)"""
    if tiles == 1:
        return reg_tree(bits, 0, cols, col_stride, True)
    if tiles == 2:
        return ('"s_bitcmp1_b32 %[idx], 2\\n\\t"\n"s_cbranch_scc1 51f\\n\\t"\n' + reg_tree(bits, 0, cols, col_stride, False) +
                '"51:\\n\\t"\n' + reg_tree(bits, 1, cols, col_stride, True))
    if tiles == 3:
        return ('"s_bitcmp1_b32 %[idx], 3\\n\\t"\n"s_cbranch_scc1 52f\\n\\t"\n"s_bitcmp1_b32 %[idx], 2\\n\\t"\n"s_cbranch_scc1 51f\\n\\t"\n' +
                reg_tree(bits, 0, cols, col_stride, False) + '"51:\\n\\t"\n' + reg_tree(bits, 1, cols, col_stride, False) +
                '"52:\\n\\t"\n' + reg_tree(bits, 2, cols, col_stride, True))
    return ('"s_bitcmp1_b32 %[idx], 3\\n\\t"\n"s_cbranch_scc1 52f\\n\\t"\n"s_bitcmp1_b32 %[idx], 2\\n\\t"\n"s_cbranch_scc1 51f\\n\\t"\n' +
            reg_tree(bits, 0, cols, col_stride, False) + '"51:\\n\\t"\n' + reg_tree(bits, 1, cols, col_stride, False) +
            '"52:\\n\\t"\n"s_bitcmp1_b32 %[idx], 2\\n\\t"\n"s_cbranch_scc1 53f\\n\\t"\n' + reg_tree(bits, 2, cols, col_stride, False) +
            '"53:\\n\\t"\n' + reg_tree(bits, 3, cols, col_stride, True))


def function(bits, tiles, cols):
    T = "double" if bits == 64 else "float"
    V = "v4d" if bits == 64 else "v4f"
    col_stride = 16 * (8 if bits == 64 else 4)  # the second tile column's 16 elements follow the first's in the LDS strip
    args = ", ".join(f"{V} &t{t}{j}" for t in range(tiles) for j in range(cols))
    ops = ", ".join(f'[a{t}{j}{r}] "+v"(t{t}{j}[{r}])' for t in range(tiles) for j in range(cols) for r in range(4))
    body = ('"s_sub_u32 %[idx], %[pos], %[base]\\n\\t"\n'
            f'"s_cmp_ge_u32 %[idx], {4 * tiles}\\n\\t"\n'  # unsigned: a slot below the block wraps around and is skipped too
            '"s_cbranch_scc1 9f\\n\\t"\n'
            '"s_and_saveexec_b64 %[save], %[mask]\\n\\t"\n' + tile_tree(bits, tiles, cols, col_stride) +
            '"8:\\n\\t"\n"s_nop 1\\n\\t"\n"s_mov_b64 exec, %[save]\\n"\n"9:"\n')
    return (f"// {tiles} tile row(s) x {cols} tile column(s), {T}\n"
            f"template <int BASE4>\n__device__ __forceinline__ void gather_tree_{cols}x{tiles}({args}, unsigned addr, int pos, unsigned long long mask)\n"
            "{\n    unsigned long long save;\n    unsigned idx;\n    asm volatile(\n" + body +
            f"        : {ops}, [save] \"=&s\"(save), [idx] \"=&s\"(idx)\n"
            "        : [addr] \"v\"(addr), [pos] \"s\"(pos), [mask] \"s\"(mask), [base] \"n\"(BASE4)\n        : \"scc\", \"memory\");\n}\n\n")


src = ("// gather_tree.inc -- GENERATED by tools/gen_gather_tree.py (edit that, not this). See its docstring.\n"
       "// Overloads on the tile type: v4d (fp64 accumulator tile, 8 VGPRs) and v4f (fp32, 4 VGPRs).\n#pragma once\n\nnamespace matinv {\n\n")
for bits in (64, 32):
    for tiles in (1, 2, 3, 4):
        src += function(bits, tiles, 1)
    for tiles in (1, 2, 3):
        src += function(bits, tiles, 2)

# one tile row, NT tile columns wide: the one-wavefront kernel of tilep_impl.hpp (NT <= 4)
for bits in (64, 32):
    for cols in (3, 4):
        src += function(bits, 1, cols)
src += "}  // namespace matinv\n"
open(OUT, "w").write(src)
print("wrote", os.path.normpath(OUT), len(src), "bytes")
