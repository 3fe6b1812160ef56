"""Generator of csrc/gemm_q4_loop.inc: the hand-placed instruction stream of gemm_nt_q4_kernel's K loop (gfx950).

Why a generator: the loop is ~200 instructions per K tile whose ORDER is the design (one memory instruction per MFMA gap,
waits and barriers at fixed MFMA indices), with physical fragment registers.  hipcc's scheduler does not keep such an order
(round 3's gemm_nt_w4_kernel: sched_group_barrier shapes, 1174 TFLOP/s where the stream below is built for >= 1400), so the
stream is emitted as ONE inline-asm statement per output tile; this script writes its text.  Run:  python tools/gen_q4_loop.py

Structure per output tile (4 waves, wave tile 128x128 as 4 x 4 v_mfma_f32_32x32x16_bf16 accumulators in AGPRs):
  tile top : vmcnt(0) | barrier | read k-half 0 of K tile 0 (16 ds_read_b128)
  body t   : 64 MFMAs (k-half 0: set 0, k-half 1: set 1), and between them
               8 reads X(t) k-half 1 | lgkm(0) barrier | 8 DMA X(t+2) interleaved with 8 reads W(t) k-half 1 | lgkm(0) barrier |
               8 DMA W(t+2) | vmcnt barrier | 8 reads X(t+1) k-half 0 | vmcnt barrier | 8 reads W(t+1) k-half 0
  The DMA stream runs two K tiles ahead in TWO LDS buffers (tile t+2 overwrites tile t's buffer operand by operand as soon
  as every wave has read it), and continues across the output-tile boundary: the last two bodies of a tile fetch K tiles 0
  and 1 of the NEXT output tile (a descriptor with num_records = 0 when there is none: the loads become no-ops).
LDS image per operand and K tile: 32 pieces of 8 rows x 128 B, each piece = one wave-wide DMA instruction (lane l -> row
l >> 3, 16-byte chunk l & 7: whole 128-byte lines from global memory, no source swizzle), pieces 1040 bytes apart: the 16
bytes of padding make the fragment reads conflict-free (row of lane r: 8 (r >> 1) + 2 f + (r & 1)).
"""
import os

PIECE = 1040
OPER = 32 * PIECE
BUF = 2 * OPER

# ---- physical registers (all listed as clobbers of the asm statement)
V_FRAG = 64            # v[64:191]: fragment sets
V_XRD, V_WRD, V_XVO, V_WVO, V_TOGX, V_TOGW = 192, 193, 194, 195, 196, 197
S_SRDX, S_SRDW = 36, 40
S_SOFFX, S_SOFFW = 44, 52
S_POS, S_DSTX, S_DSTW, S_TOGX, S_TOGW, S_NK, S_MID, S_T0, S_T1 = 60, 61, 62, 63, 64, 65, 66, 67, 68
S_XN, S_WN, S_NREC = 70, 72, 74          # s[70:71] next X base, s[72:73] next W base, s74 next num_records
V_CLOBBER = list(range(V_FRAG, V_TOGW + 1))
S_CLOBBER = list(range(36, 76))

# ---- operand numbers of the main statement
# accumulators are PHYSICAL: acc[i][j] = a[16 (4 i + j) : + 15] (clobbers; the epilogue reads them with v_accvgpr_read)
OP_XRD, OP_WRD, OP_XVO, OP_WVO = 0, 1, 2, 3          # "v"
OP_XCUR, OP_WCUR, OP_XNEXT, OP_WNEXT = 4, 5, 6, 7    # "s", 64 bit
OP_LDA32, OP_LDB32, OP_NKF, OP_DSTW = 8, 9, 10, 11   # "s"
# the instrumented statement (timing probe) has two 64-bit "=&s" outputs in front: %0 = s_memtime behind the tile-top
# barrier, %1 = s_memtime behind the last MFMA; its inputs are the ones above shifted by 2


def frag(setn, op, s, f):
    b = V_FRAG + setn * 64 + op * 32 + s * 16 + f * 4
    return 'v[%d:%d]' % (b, b + 3)


def acc(i, j):
    b = 16 * (4 * i + j)
    return 'a[%d:%d]' % (b, b + 15)


def mfma(kh, q, zero_c):
    s, j, i = q // 16, (q % 16) // 4, q % 4
    c = '0' if zero_c else acc(i, j)
    return 'v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s' % (acc(i, j), frag(kh, 1, s, j), frag(kh, 0, s, i), c)


def read(op, setn, kh, n):
    """n-th (0..7) fragment read of operand op for k-half kh into register set setn"""
    s, f = n // 4, n % 4
    return 'ds_read_b128 %s, v%d offset:%d' % (frag(setn, op, s, f), V_XRD if op == 0 else V_WRD, f * 256 + (2 * kh + s) * 32)


def dma(op, j):
    srd = S_SRDX if op == 0 else S_SRDW
    soff = (S_SOFFX if op == 0 else S_SOFFW) + j
    return ['buffer_load_dwordx4 v%d, s[%d:%d], s%d offen lds' % (V_XVO if op == 0 else V_WVO, srd, srd + 3, soff),
            's_add_u32 m0, m0, %d' % (4 * PIECE)]


def srd_setup(xop, wop):
    o = []
    o += ['s_mov_b64 s[%d:%d], %%%d' % (S_SRDX, S_SRDX + 1, xop), 's_mov_b64 s[%d:%d], %%%d' % (S_SRDW, S_SRDW + 1, wop)]
    for b in (S_SRDX, S_SRDW):
        o += ['s_and_b32 s%d, s%d, 0xffff' % (b + 1, b + 1), 's_mov_b32 s%d, 0x80000000' % (b + 2), 's_mov_b32 s%d, 0x00020000' % (b + 3)]
    return o


def soff_setup(lda_op, ldb_op):
    o = []
    for j in range(8):
        o += ['s_mul_i32 s%d, %%%d, %d' % (S_SOFFX + j, lda_op, j), 's_mul_i32 s%d, %%%d, %d' % (S_SOFFW + j, ldb_op, j)]
    return o


# Schedule variants (positions = "after MFMA p" of the 64 MFMAs of a body).  The kernel is instantiated once per variant;
# TELL_Q4_VAR picks one at run time (A/B inside one process, same box).
#   0: X reads 0-7 | lgkm(0) barrier 9 | DMA X / W reads interleaved 11-23 | lgkm(0) barrier 25 | ...
#   1: X reads 0-7, W reads 8-15 back to back | lgkm(8) barrier 12 (X done: LDS returns in order) | DMA X 14-28 |
#      lgkm(0) barrier 19 | DMA W from 30: the waits sit >= 4 MFMAs behind the reads they cover
#   2: variant 0 with both read waits one MFMA later (10 / 26)
VARIANTS = (0, 1, 2)


def body(first, last, var=0, extra=None):
    ev = {}

    def at(p, *ins):
        ev.setdefault(p, []).extend(ins)

    xorx = 'v_xor_b32 v%d, v%d, v%d' % (V_XRD, V_TOGX, V_XRD)
    xorw = 'v_xor_b32 v%d, v%d, v%d' % (V_WRD, V_TOGW, V_WRD)
    if var == 1:
        for n in range(8):
            at(n, read(0, 1, 1, n))
        at(7, xorx)
        for n in range(8):
            at(8 + n, read(1, 1, 1, n))
        at(15, xorw)
        at(12, 's_waitcnt lgkmcnt(5)', 's_barrier')                # issued so far: 8 X + 5 W reads -> the X reads are done
        at(13, 's_mov_b32 m0, s%d' % S_DSTX, 's_nop 0')
        for n, p in enumerate((14, 16, 18, 20, 22, 24, 26)):
            at(p, *dma(0, n))
        at(19, 's_waitcnt lgkmcnt(0)', 's_barrier')
        at(28, dma(0, 7)[0], 's_mov_b32 m0, s%d' % S_DSTW)
        at(30, *dma(1, 0))
        at(31, *dma(1, 1))
    else:
        w1, w2 = (9, 25) if var == 0 else (10, 26)
        for n in range(8):
            at(n, read(0, 1, 1, n))
        at(7, xorx)
        at(w1, 's_waitcnt lgkmcnt(0)', 's_barrier')
        at(w1 + 1, 's_mov_b32 m0, s%d' % S_DSTX, 's_nop 0')
        for n, p in enumerate((w1 + 2, w1 + 4, w1 + 6, w1 + 8, w1 + 10)):
            at(p, *dma(0, n))
        for n, p in enumerate((12, 14, 16, 18, 20, 21, 22, 23)):
            at(p, read(1, 1, 1, n))
        at(23, xorw)
        at(w2, 's_waitcnt lgkmcnt(0)', 's_barrier')
        at(w2 + 1, *dma(0, 5))
        at(w2 + 2, *dma(0, 6))
        at(w2 + 3, dma(0, 7)[0], 's_mov_b32 m0, s%d' % S_DSTW)
        at(30 if var == 0 else 31, *dma(1, 0))
        at(31 if var == 0 else 32, *dma(1, 1))
    if not last:
        at(33, 's_waitcnt vmcnt(18)', 's_barrier')
        for n in range(8):
            at(34 + n, read(0, 0, 0, n))
    for n, p in enumerate((42, 44, 46, 48, 50)):
        at(p, *dma(1, 2 + n))
    if not last:
        at(51, 's_waitcnt vmcnt(15)', 's_barrier')
        for n in range(8):
            at(52 + n, read(1, 0, 0, n))
    at(60, dma(1, 7)[0])
    # stream bookkeeping: the descriptors move to the next stream position; at position nk they jump to the next output tile
    at(61, 's_add_u32 s%d, s%d, 128' % (S_SRDX, S_SRDX), 's_addc_u32 s%d, s%d, 0' % (S_SRDX + 1, S_SRDX + 1),
       's_add_u32 s%d, s%d, 128' % (S_SRDW, S_SRDW), 's_addc_u32 s%d, s%d, 0' % (S_SRDW + 1, S_SRDW + 1),
       's_add_u32 s%d, s%d, 1' % (S_POS, S_POS))
    at(62, 's_cmp_eq_u32 s%d, s%d' % (S_POS, S_NK),
       's_cselect_b32 s%d, s%d, s%d' % (S_SRDX, S_XN, S_SRDX), 's_cselect_b32 s%d, s%d, s%d' % (S_SRDX + 1, S_XN + 1, S_SRDX + 1),
       's_cselect_b32 s%d, s%d, s%d' % (S_SRDX + 2, S_NREC, S_SRDX + 2),
       's_cselect_b32 s%d, s%d, s%d' % (S_SRDW, S_WN, S_SRDW), 's_cselect_b32 s%d, s%d, s%d' % (S_SRDW + 1, S_WN + 1, S_SRDW + 1),
       's_cselect_b32 s%d, s%d, s%d' % (S_SRDW + 2, S_NREC, S_SRDW + 2))
    at(63, 's_xor_b32 s%d, s%d, s%d' % (S_DSTX, S_DSTX, S_TOGX), 's_xor_b32 s%d, s%d, s%d' % (S_DSTW, S_DSTW, S_TOGW))
    for p_, ins in (extra or {}).items():                   # (gen_q4e_loop.py: deferred stores, the bias DMA)
        ev.setdefault(p_, []).extend(ins)
    out = ['s_waitcnt lgkmcnt(0)']
    for m in range(64):
        kh, q = m // 32, m % 32
        out.append(mfma(kh, q, first and kh == 0 and q < 16))
        out += ev.get(m, [])
    return out


def main_statement(var=0, dbg=False):
    text = _main_statement(var, dbg)
    if dbg:                                                 # shift the input operand numbers behind the two outputs
        import re
        text = [re.sub(r'%(\d+)', lambda m: '%%%d' % (int(m.group(1)) + 2), ln) if not ln.startswith('s_memtime') else ln for ln in text]
    return text


def _main_statement(var=0, dbg=False):
    o = []
    # ---- tile top: state into physical registers
    o += ['v_mov_b32 v%d, %%%d' % (V_XRD, OP_XRD), 'v_mov_b32 v%d, %%%d' % (V_WRD, OP_WRD),
          'v_mov_b32 v%d, %%%d' % (V_XVO, OP_XVO), 'v_mov_b32 v%d, %%%d' % (V_WVO, OP_WVO)]
    # per-lane XOR masks between the two buffers' read addresses
    o += ['v_add_u32 v%d, %d, v%d' % (V_TOGX, BUF, V_XRD), 'v_xor_b32 v%d, v%d, v%d' % (V_TOGX, V_TOGX, V_XRD),
          'v_add_u32 v%d, %d, v%d' % (V_TOGW, BUF, V_WRD), 'v_xor_b32 v%d, v%d, v%d' % (V_TOGW, V_TOGW, V_WRD)]
    o += srd_setup(OP_XCUR, OP_WCUR)
    o += soff_setup(OP_LDA32, OP_LDB32)
    o += ['s_mov_b64 s[%d:%d], %%%d' % (S_XN, S_XN + 1, OP_XNEXT), 's_mov_b64 s[%d:%d], %%%d' % (S_WN, S_WN + 1, OP_WNEXT),
          's_and_b32 s%d, s%d, 0xffff' % (S_XN + 1, S_XN + 1), 's_and_b32 s%d, s%d, 0xffff' % (S_WN + 1, S_WN + 1),
          's_and_b32 s%d, %%%d, 0xffff' % (S_NK, OP_NKF),                      # nk
          's_lshr_b32 s%d, %%%d, 16' % (S_T0, OP_NKF),                         # has_next (0 / 1)
          's_lshl_b32 s%d, s%d, 31' % (S_NREC, S_T0),                          # num_records of the next tile: 0 or 2^31
          's_mov_b32 s%d, %%%d' % (S_DSTX, OP_DSTW), 's_add_u32 s%d, s%d, %d' % (S_DSTW, S_DSTX, OPER),
          's_add_u32 s%d, s%d, %d' % (S_T0, S_DSTX, BUF), 's_xor_b32 s%d, s%d, s%d' % (S_TOGX, S_T0, S_DSTX),
          's_add_u32 s%d, s%d, %d' % (S_T0, S_DSTW, BUF), 's_xor_b32 s%d, s%d, s%d' % (S_TOGW, S_T0, S_DSTW),
          's_sub_u32 s%d, s%d, 2' % (S_MID, S_NK),
          's_mov_b32 s%d, 2' % S_POS]
    # stream position 2 may already be the next tile (nk == 2)
    o += ['s_cmp_eq_u32 s%d, s%d' % (S_POS, S_NK),
          's_cselect_b32 s%d, s%d, s%d' % (S_SRDX, S_XN, S_SRDX), 's_cselect_b32 s%d, s%d, s%d' % (S_SRDX + 1, S_XN + 1, S_SRDX + 1),
          's_cselect_b32 s%d, s%d, s%d' % (S_SRDX + 2, S_NREC, S_SRDX + 2),
          's_cselect_b32 s%d, s%d, s%d' % (S_SRDW, S_WN, S_SRDW), 's_cselect_b32 s%d, s%d, s%d' % (S_SRDW + 1, S_WN + 1, S_SRDW + 1),
          's_cselect_b32 s%d, s%d, s%d' % (S_SRDW + 2, S_NREC, S_SRDW + 2)]
    # K tiles 0 and 1 of this output tile were put in flight by the previous statement: wait (this also retires the
    # previous epilogue's stores, which share the counter), then read k-half 0 of K tile 0
    o += ['s_waitcnt vmcnt(0)', 's_barrier']
    if dbg:
        o.append('s_memtime %0')
    for n in range(8):
        o.append(read(0, 0, 0, n))
    for n in range(8):
        o.append(read(1, 0, 0, n))
    o += body(True, False, var)
    o += ['s_cmp_eq_u32 s%d, 0' % S_MID, 's_cbranch_scc1 2f', '.p2align 6', '1:']
    o += body(False, False, var)
    o += ['s_sub_u32 s%d, s%d, 1' % (S_MID, S_MID), 's_cmp_eq_u32 s%d, 0' % S_MID, 's_cbranch_scc0 1b', '2:']
    o += body(False, True, var)
    if dbg:
        o += ['s_memtime %1', 's_waitcnt lgkmcnt(0)']
    o += ['s_nop 15', 's_nop 15']
    return o


def prologue_statement():
    """K tiles 0 and 1 of a workgroup's FIRST output tile.  Operands: %0 xvoff %1 wvoff (v), %2 X %3 W (s64), %4 lda32 %5 ldb32 %6 dstw (s)"""
    o = ['s_mov_b64 s[%d:%d], %%2' % (S_SRDX, S_SRDX + 1), 's_mov_b64 s[%d:%d], %%3' % (S_SRDW, S_SRDW + 1)]
    for b in (S_SRDX, S_SRDW):
        o += ['s_and_b32 s%d, s%d, 0xffff' % (b + 1, b + 1), 's_mov_b32 s%d, 0x80000000' % (b + 2), 's_mov_b32 s%d, 0x00020000' % (b + 3)]
    o += soff_setup(4, 5)
    for kt in range(2):
        for op in range(2):
            o += ['s_add_u32 m0, %%6, %d' % (kt * BUF + op * OPER), 's_nop 0']
            for j in range(8):
                srd = S_SRDX if op == 0 else S_SRDW
                o += ['buffer_load_dwordx4 %%%d, s[%d:%d], s%d offen lds' % (op, srd, srd + 3, (S_SOFFX if op == 0 else S_SOFFW) + j),
                      's_add_u32 m0, m0, %d' % (4 * PIECE)]
        if kt == 0:
            o += ['s_add_u32 s%d, s%d, 128' % (S_SRDX, S_SRDX), 's_addc_u32 s%d, s%d, 0' % (S_SRDX + 1, S_SRDX + 1),
                  's_add_u32 s%d, s%d, 128' % (S_SRDW, S_SRDW), 's_addc_u32 s%d, s%d, 0' % (S_SRDW + 1, S_SRDW + 1)]
    return o


def c_string(lines):
    return '\n'.join('  "%s\\n\\t"' % ln for ln in lines)


def clobbers(v, s, a=()):
    return ', '.join(['"memory"', '"scc"'] + ['"v%d"' % r for r in v] + ['"s%d"' % r for r in s] + ['"a%d"' % r for r in a])


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, '..', 'transform-and-tell_amd', 'csrc', 'gemm_q4_loop.inc')
    with open(path, 'w') as f:
        f.write('// GENERATED by tools/gen_q4_loop.py - do not edit.  The K loop of gemm_nt_q4_kernel (csrc/gemm_q4.hip) as inline-asm text.\n')
        f.write('#define Q4_PIECE %d\n#define Q4_OPER %d\n#define Q4_BUF %d\n' % (PIECE, OPER, BUF))
        for var in VARIANTS:
            f.write('#define Q4_MAIN_ASM_%d \\\n' % var + c_string(main_statement(var)).replace('\n', ' \\\n') + '\n')
        f.write('#define Q4_MAIN_ASM_DBG \\\n' + c_string(main_statement(0, True)).replace('\n', ' \\\n') + '\n')
        f.write('#define Q4_N_VARIANTS %d\n' % len(VARIANTS))
        f.write('#define Q4_MAIN_CLOBBERS ' + clobbers(V_CLOBBER, S_CLOBBER, range(256)) + '\n')
        f.write('#define Q4_PROLOGUE_ASM \\\n' + c_string(prologue_statement()).replace('\n', ' \\\n') + '\n')
        f.write('#define Q4_PROLOGUE_CLOBBERS ' + clobbers([], list(range(36, 60))) + '\n')
    print('wrote', os.path.normpath(path))


if __name__ == '__main__':
    main()
