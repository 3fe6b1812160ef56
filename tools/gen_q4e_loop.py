"""Generator of csrc/gemm_q4e_loop.inc: gemm_nt_q4_kernel's K loop (tools/gen_q4_loop.py - read that first) WITH THE EPILOGUE
OF THE PREVIOUS OUTPUT TILE INSIDE IT (gemm_nt_q4e_kernel, csrc/gemm_q4e.hip).

Measured on gemm_nt_q4_kernel (tools/probes/q4_variants.py, s_memtime): the K loop runs at the MFMA floor (2080 clk per K tile
of 64 MFMAs), but every output tile then spends 10.6 k clk in its epilogue (16.5 k with GELU) against 33 k of K loop at
K = 1024 - matrix cores idle, and all 256 CUs write their 128 KB at the same moment (a 32 MB burst at ~7.8 TB/s: the stores
cost the same whether they cover whole lines or not).  Here the tile's statement BEGINS with the drain of the previous tile's
accumulators (v_accvgpr_read -> acc * alpha + bias -> activation -> bf16 pairs: VALU work that cannot be avoided, ~2.5 k clk)
and only a quarter of the stores; the other 24 stores of a lane stay in 96 VGPRs and leave one by one from the first bodies
of the K loop, behind the second counted wait of a body (the spot with the longest distance to the next wait that a store
acknowledgement could delay).  Stores and loads share vmcnt; extra outstanding stores only make a counted wait stricter,
never unsafe (loads return in order among loads).
The bias of the CURRENT tile is fetched by LDS-DMA (256 floats, 64 per wave) into one of two 1 KB slots behind the operand
buffers in the first body and read by the NEXT statement's drain: no global-load latency in the drain.
Register map: compiler v0-19 | state v20-31 | packed outputs P = v[32:127] | fragments v[128:255] (the drain's temporaries T =
set 0, bias B = set 1: dead between two tiles).  The last tile of a workgroup is drained by the FLUSH statement."""
import os
import gen_q4_loop as g

PIECE, OPER, BUF = g.PIECE, g.OPER, g.BUF
g.V_FRAG = 128
g.V_XRD, g.V_WRD, g.V_XVO, g.V_WVO, g.V_TOGX, g.V_TOGW = 20, 21, 22, 23, 24, 25
V_COFF, V_BRD, V_BVO = 26, 27, 28
V_P, V_T, V_B = 32, 128, 192
V_CLOBBER = list(range(20, 256))
S_SRDC, S_CI, S_ACT, S_ALPHA, S_BDST, S_SRDB = 76, 80, 83, 84, 86, 88          # s[76:79] C descriptor, s80-82 row offsets of i = 1..3
S_CONST = dict(RS2=44, A6=46, A5=48, A4=50, A3=52, A2=54, A1=56, CM=58)   # GELU constants: free while no DMA offsets are needed
S_CLOBBER = list(range(36, 100))
# How the 24 deferred 16-byte stores of a lane leave.  nsb: bodies that carry stores; pos: positions inside a body ("after MFMA
# p"; only the window behind the second counted wait, p = 52..63, is far enough from the next strict wait for a store's
# acknowledgement not to delay it); width: 4 = dwordx4, 2 = two dwordx2 per 16 bytes; late_wait: the wait for this tile's first
# operands (and the barrier) sits BEHIND the drain, which does not need them; no_imm: timing probe (wrong results) without the
# drain's 8 immediate stores.  PROBES are instantiated with s_memtime stamps for tools/probes/q4_variants.py (TELL_Q4E_VAR).
PROD = dict(nsb=12, pos=(53, 61), width=4, late_wait=True, no_imm=False)      # measured best of PROBES (tools/probes/q4_variants.py)
# The production configuration with stamps (TELL_Q4E_VAR=0).  Round 4 compared six configurations here (nsb 8 / 12, store
# positions, dwordx2 pairs, the late wait, no immediate stores: profiles/r04_q4_variants.txt has their numbers); the losing
# five were removed from the shipped library in round 5 - put their dicts back here to re-measure them.
PROBES = (dict(PROD),)

# operands of the main statement
OP = dict(XRD=0, WRD=1, XVO=2, WVO=3, COFF=4, BRD=5, BVO=6, XCUR=7, WCUR=8, XNEXT=9, WNEXT=10, LDA32=11, LDB32=12, NKF=13,
          DSTW=14, CPREV=15, BIAS=16, LDC2=17, ALPHA=18, FLAGS=19, BDST=20)


def fbits(x):
    import struct
    return '0x%08x' % struct.unpack('<I', struct.pack('<f', x))[0]


def col_off(q):                       # byte offset of column group q (8 bf16) inside the lane's 64 columns
    return 32 * (q & 1) + 64 * (q >> 1)


def const_setup():
    vals = dict(RS2=0.70710678118654752, A6=0.0000430638, A5=0.0002765672, A4=0.0001520143, A3=0.0092705272, A2=0.0422820123,
                A1=0.0705230784, CM=-0.70710678118654752)
    return ['s_mov_b32 s%d, %s' % (S_CONST[k], fbits(v)) for k, v in vals.items()]


def gelu(t0, x0, q0, npairs=8):
    """exact-erf GELU (gemm_common.h, Abramowitz-Stegun 7.1.28) in place on 2 npairs values at v[t0..]; temporaries x0.., q0..
    (2 npairs registers each).  Instruction k of every pair before instruction k + 1 of any: dependent packed operations
    sit npairs issues apart."""
    def pr(b, n):
        return 'v[%d:%d]' % (b + 2 * n, b + 2 * n + 1)

    def sc(k):
        return 's[%d:%d]' % (S_CONST[k], S_CONST[k] + 1)
    o = []
    R = range(npairs)
    for n in R:
        o += ['v_and_b32 v%d, 0x7fffffff, v%d' % (x0 + 2 * n, t0 + 2 * n), 'v_and_b32 v%d, 0x7fffffff, v%d' % (x0 + 2 * n + 1, t0 + 2 * n + 1)]
    o += ['v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]' % (pr(x0, n), pr(x0, n), sc('RS2')) for n in R]
    o += ['v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]' % (pr(q0, n), pr(x0, n), sc('A6')) for n in R]
    o += ['v_pk_add_f32 %s, %s, %s op_sel_hi:[1,0]' % (pr(q0, n), pr(q0, n), sc('A5')) for n in R]
    for k in ('A4', 'A3', 'A2', 'A1'):
        o += ['v_pk_fma_f32 %s, %s, %s, %s op_sel_hi:[1,1,0]' % (pr(q0, n), pr(q0, n), pr(x0, n), sc(k)) for n in R]
    o += ['v_pk_fma_f32 %s, %s, %s, 1.0 op_sel_hi:[1,1,0]' % (pr(q0, n), pr(q0, n), pr(x0, n)) for n in R]
    for _ in range(4):
        o += ['v_pk_mul_f32 %s, %s, %s' % (pr(q0, n), pr(q0, n), pr(q0, n)) for n in R]
    for n in R:
        o += ['v_rcp_f32 v%d, v%d' % (q0 + 2 * n, q0 + 2 * n), 'v_rcp_f32 v%d, v%d' % (q0 + 2 * n + 1, q0 + 2 * n + 1)]
    o += ['v_pk_mul_f32 %s, %s, %s op_sel_hi:[1,0]' % (pr(x0, n), pr(x0, n), sc('CM')) for n in R]          # -|v| / 2
    for n in R:
        o += ['v_max_f32 v%d, 0, v%d' % (t0 + 2 * n, t0 + 2 * n), 'v_max_f32 v%d, 0, v%d' % (t0 + 2 * n + 1, t0 + 2 * n + 1)]
    o += ['v_pk_fma_f32 %s, %s, %s, %s' % (pr(t0, n), pr(x0, n), pr(q0, n), pr(t0, n)) for n in R]
    return o


def store(data, i, q, width=4, half=0):
    soff = '0' if i == 0 else 's%d' % (S_CI + i - 1)
    if width == 2:
        return 'buffer_store_dwordx2 v[%d:%d], v%d, s[%d:%d], %s offen offset:%d' % (data + 2 * half, data + 2 * half + 1, V_COFF, S_SRDC, S_SRDC + 3,
                                                                                    soff, col_off(q) + 8 * half)
    return 'buffer_store_dwordx4 v[%d:%d], v%d, s[%d:%d], %s offen offset:%d' % (data, data + 3, V_COFF, S_SRDC, S_SRDC + 3, soff, col_off(q))


def drain_i(i, immediate, act, no_imm=False):
    """X fragment row i of the previous tile: T[8 q + 2 j + t] <- acc[i][j][2 q + t]; act(T * alpha + B); bf16 pairs -> T in
    place + 8 stores (immediate) or P[32 (i - 1) + 4 q + j] (deferred)."""
    o = []
    for j in range(4):
        for e in range(16):
            o.append('v_accvgpr_read_b32 v%d, a%d' % (V_T + 8 * (e >> 1) + 2 * j + (e & 1), 16 * (4 * i + j) + e))
    for n in range(32):
        o.append('v_pk_fma_f32 v[%d:%d], v[%d:%d], s[%d:%d], v[%d:%d] op_sel_hi:[1,0,1]'
                 % (V_T + 2 * n, V_T + 2 * n + 1, V_T + 2 * n, V_T + 2 * n + 1, S_ALPHA, S_ALPHA + 1, V_B + 2 * n, V_B + 2 * n + 1))
    # activation (one statement text per activation: 0 none, 1 relu, 2 gelu)
    tmp = V_B if i == 3 else V_P + 64                      # 32 temporaries: the part of P filled last, or the bias (dead after i = 3's fma)
    if act == 1:
        o += ['v_max_f32 v%d, 0, v%d' % (V_T + n, V_T + n) for n in range(64)]
    elif act == 2:
        for c in range(4):
            o += gelu(V_T + 16 * c, tmp, tmp + 16)
    for q in range(8):
        for j in range(4):
            dst = V_T + 8 * q + j if immediate else V_P + 32 * (i - 1) + 4 * q + j
            o.append('v_cvt_pk_bf16_f32 v%d, v%d, v%d' % (dst, V_T + 8 * q + 2 * j, V_T + 8 * q + 2 * j + 1))
    if immediate and not no_imm:
        o += [store(V_T + 8 * q, i, q) for q in range(8)]
        o.append('s_nop 1')
    return o


def drain(all_immediate, act, no_imm=False):
    o = []
    for q in range(8):                                      # B[8 q + k] = bias of column group q (LDS slot written a tile ago)
        o += ['ds_read_b128 v[%d:%d], v%d offset:%d' % (V_B + 8 * q, V_B + 8 * q + 3, V_BRD, 2 * col_off(q)),
              'ds_read_b128 v[%d:%d], v%d offset:%d' % (V_B + 8 * q + 4, V_B + 8 * q + 7, V_BRD, 2 * col_off(q) + 16)]
    o.append('s_waitcnt lgkmcnt(0)')
    o += ['v_pk_mul_f32 v[%d:%d], v[%d:%d], s[%d:%d] op_sel_hi:[1,0]' % (V_B + 2 * n, V_B + 2 * n + 1, V_B + 2 * n, V_B + 2 * n + 1, S_ALPHA, S_ALPHA + 1)
          for n in range(32)]
    for i in range(4):
        o += drain_i(i, all_immediate or i == 0, act, no_imm)
    return o


def c_setup(cprev_op, ldc2_op, alpha_op, hasprev_bit_from=None):
    o = ['s_mov_b64 s[%d:%d], %%%d' % (S_SRDC, S_SRDC + 1, cprev_op), 's_and_b32 s%d, s%d, 0xffff' % (S_SRDC + 1, S_SRDC + 1),
         's_mov_b32 s%d, 0x00020000' % (S_SRDC + 3),
         's_mov_b32 s%d, %%%d' % (S_CI, ldc2_op), 's_lshl_b32 s%d, %%%d, 1' % (S_CI + 1, ldc2_op), 's_mul_i32 s%d, %%%d, 3' % (S_CI + 2, ldc2_op),
         's_mov_b32 s%d, %%%d' % (S_ALPHA, alpha_op)]
    return o


def main_statement(act, cfg=None, dbg=False):
    text = _main_statement(act, cfg or PROD, dbg)
    if dbg:                                                 # three 64-bit "=&s" outputs in front: inputs shift by 3
        import re
        text = [re.sub(r'%(\d+)', lambda m: '%%%d' % (int(m.group(1)) + 3), ln) if not ln.startswith('s_memtime') else ln for ln in text]
    return text


def _main_statement(act, cfg, dbg=False):
    o = []
    var = 0
    o += ['v_mov_b32 v%d, %%%d' % (g.V_XRD, OP['XRD']), 'v_mov_b32 v%d, %%%d' % (g.V_WRD, OP['WRD']),
          'v_mov_b32 v%d, %%%d' % (g.V_XVO, OP['XVO']), 'v_mov_b32 v%d, %%%d' % (g.V_WVO, OP['WVO']),
          'v_mov_b32 v%d, %%%d' % (V_COFF, OP['COFF']), 'v_mov_b32 v%d, %%%d' % (V_BRD, OP['BRD']), 'v_mov_b32 v%d, %%%d' % (V_BVO, OP['BVO'])]
    o += ['v_add_u32 v%d, %d, v%d' % (g.V_TOGX, BUF, g.V_XRD), 'v_xor_b32 v%d, v%d, v%d' % (g.V_TOGX, g.V_TOGX, g.V_XRD),
          'v_add_u32 v%d, %d, v%d' % (g.V_TOGW, BUF, g.V_WRD), 'v_xor_b32 v%d, v%d, v%d' % (g.V_TOGW, g.V_TOGW, g.V_WRD)]
    o += g.srd_setup(OP['XCUR'], OP['WCUR'])
    o += c_setup(OP['CPREV'], OP['LDC2'], OP['ALPHA'])
    o += ['s_lshr_b32 s%d, %%%d, 8' % (g.S_T0, OP['FLAGS']), 's_lshl_b32 s%d, s%d, 31' % (S_SRDC + 2, g.S_T0),   # C num_records: 0 without a previous tile
          's_mov_b64 s[%d:%d], %%%d' % (S_SRDB, S_SRDB + 1, OP['BIAS']), 's_and_b32 s%d, s%d, 0xffff' % (S_SRDB + 1, S_SRDB + 1),
          's_mov_b32 s%d, 1024' % (S_SRDB + 2), 's_mov_b32 s%d, 0x00020000' % (S_SRDB + 3),
          's_mov_b32 s%d, %%%d' % (S_BDST, OP['BDST'])]
    o += const_setup()
    # K tiles 0 / 1 of this tile were put in flight by the previous statement's last two bodies.  The drain needs neither them
    # nor a barrier (the previous tile's bias slot was written by DMAs that every wave has waited for, bodies ago)
    if not cfg['late_wait']:
        o += ['s_waitcnt vmcnt(0)', 's_barrier']
    if dbg:
        o += ['s_memtime %0']
    o += ['s_cmp_eq_u32 s%d, 0' % (S_SRDC + 2), 's_cbranch_scc1 90f']       # no previous tile: nothing to drain
    o += drain(False, act, cfg['no_imm'])
    o += ['90:']
    if cfg['late_wait']:
        o += ['s_waitcnt vmcnt(0)', 's_barrier']
    if dbg:
        o += ['s_memtime %1']
    o += g.soff_setup(OP['LDA32'], OP['LDB32'])
    o += ['s_mov_b64 s[%d:%d], %%%d' % (g.S_XN, g.S_XN + 1, OP['XNEXT']), 's_mov_b64 s[%d:%d], %%%d' % (g.S_WN, g.S_WN + 1, OP['WNEXT']),
          's_and_b32 s%d, s%d, 0xffff' % (g.S_XN + 1, g.S_XN + 1), 's_and_b32 s%d, s%d, 0xffff' % (g.S_WN + 1, g.S_WN + 1),
          's_and_b32 s%d, %%%d, 0xffff' % (g.S_NK, OP['NKF']),
          's_lshr_b32 s%d, %%%d, 16' % (g.S_T0, OP['NKF']), 's_lshl_b32 s%d, s%d, 31' % (g.S_NREC, g.S_T0),
          's_mov_b32 s%d, %%%d' % (g.S_DSTX, OP['DSTW']), 's_add_u32 s%d, s%d, %d' % (g.S_DSTW, g.S_DSTX, OPER),
          's_add_u32 s%d, s%d, %d' % (g.S_T0, g.S_DSTX, BUF), 's_xor_b32 s%d, s%d, s%d' % (g.S_TOGX, g.S_T0, g.S_DSTX),
          's_add_u32 s%d, s%d, %d' % (g.S_T0, g.S_DSTW, BUF), 's_xor_b32 s%d, s%d, s%d' % (g.S_TOGW, g.S_T0, g.S_DSTW),
          's_sub_u32 s%d, s%d, %d' % (g.S_MID, g.S_NK, cfg['nsb'] + 1),
          's_mov_b32 s%d, 2' % g.S_POS]
    for n in range(8):
        o.append(g.read(0, 0, 0, n))
    for n in range(8):
        o.append(g.read(1, 0, 0, n))
    pend = []                                                # the deferred stores, in issue order
    for gi in range(24):
        i, q = 1 + gi // 8, gi % 8
        for half in range(4 // cfg['width']):
            pend.append(store(V_P + 32 * (i - 1) + 4 * q, i, q, cfg['width'], half))
    assert len(pend) <= cfg['nsb'] * len(cfg['pos'])
    for b in range(cfg['nsb']):
        extra = {}
        for pos in cfg['pos']:
            if pend:
                extra.setdefault(pos, []).append(pend.pop(0))
        if b == 0:                                           # this tile's bias -> its LDS slot (read by the next statement's drain)
            extra.setdefault(8, []).extend(['s_mov_b32 m0, s%d' % S_BDST, 's_nop 0',
                                            'buffer_load_dword v%d, s[%d:%d], 0 offen lds' % (V_BVO, S_SRDB, S_SRDB + 3)])
        o += g.body(b == 0, False, var, extra)
    o += ['s_cmp_eq_u32 s%d, 0' % g.S_MID, 's_cbranch_scc1 2f', '.p2align 6', '1:']
    o += g.body(False, False, var)
    o += ['s_sub_u32 s%d, s%d, 1' % (g.S_MID, g.S_MID), 's_cmp_eq_u32 s%d, 0' % g.S_MID, 's_cbranch_scc0 1b', '2:']
    o += g.body(False, True, var)
    if dbg:
        o += ['s_memtime %2', 's_waitcnt lgkmcnt(0)']
    o += ['s_nop 15', 's_nop 15']
    return o


def flush_statement(act):
    """The workgroup's last tile.  Operands: %0 coff %1 brd (v), %2 C tile (s64), %3 ldc2 %4 alpha (s)"""
    o = ['v_mov_b32 v%d, %%0' % V_COFF, 'v_mov_b32 v%d, %%1' % V_BRD]
    o += c_setup(2, 3, 4)
    o += ['s_mov_b32 s%d, 0x80000000' % (S_SRDC + 2)]
    o += const_setup()
    o += ['s_waitcnt vmcnt(0)', 's_barrier']
    o += drain(True, act)
    return o


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    path = os.path.join(here, '..', 'transform-and-tell_amd', 'csrc', 'gemm_q4e_loop.inc')
    with open(path, 'w') as f:
        f.write('// GENERATED by tools/gen_q4e_loop.py - do not edit.  K loop + previous tile\'s epilogue of gemm_nt_q4e_kernel (csrc/gemm_q4e.hip).\n')
        f.write('#define Q4E_NSB %d\n' % PROD['nsb'])
        for act in range(3):
            f.write('#define Q4E_MAIN_ASM_ACT%d \\\n' % act + g.c_string(main_statement(act)).replace('\n', ' \\\n') + '\n')
            f.write('#define Q4E_FLUSH_ASM_ACT%d \\\n' % act + g.c_string(flush_statement(act)).replace('\n', ' \\\n') + '\n')
        for k, cfg in enumerate(PROBES):
            f.write('#define Q4E_PROBE_ASM_V%d \\\n' % k + g.c_string(main_statement(0, cfg, True)).replace('\n', ' \\\n') + '\n')
            f.write('#define Q4E_PROBE_NSB_V%d %d\n' % (k, cfg['nsb']))
        f.write('#define Q4E_N_PROBES %d\n' % len(PROBES))
        f.write('#define Q4E_CLOBBERS ' + g.clobbers(V_CLOBBER, S_CLOBBER, range(256)) + '\n')
    print('wrote', os.path.normpath(path))


if __name__ == '__main__':
    main()
