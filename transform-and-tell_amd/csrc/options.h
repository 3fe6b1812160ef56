// Run-time options of libtell_hip.so (tell_set_option / tell_get_option, include/tell_hip.h).
//
// Every dispatch choice a caller may steer is ONE named integer in this table, set explicitly through the C ABI - the
// library reads no environment variable.  (The Python host mirror translates TELL_<KEY> variables into tell_set_option
// calls once, when it loads the library: transform-and-tell_amd/hip.py.)  A launcher reads its options with tell_opt(),
// a relaxed load of a process-global atomic: the value in force is the one set before the launch is ISSUED (a launch
// recorded into a hipGraph keeps the choice it was recorded with).
//
// Timing probes that produce WRONG results (ablations: no epilogue, no global stores, ...) are compiled only into the
// probe build (-DTELL_PROBES -> libtell_hip_probes.so, used by tools/probes/): in the shipped library their keys do not
// exist (tell_set_option returns an error), tell_probe() is the constant 0 and the ablated kernels are not instantiated.
#pragma once
#include <atomic>

// X(enumerator, key, default)
#define TELL_OPTION_LIST(X)                                                                                             \
  /* ---- the 256x256 GEMM family (gemm.hip, gemm_q4.hip, gemm_q4e.hip, gemm_pp2.hip) */                                \
  X(OPT_GEMM_Q4, "gemm_q4", 1)             /* 0: the ping-pong kernels take the whole-tile launches */                  \
  X(OPT_GEMM_Q4E, "gemm_q4e", 1)           /* 0: plain q4; 2: the in-loop epilogue wherever it applies */              \
  X(OPT_Q4_PARTIAL, "q4_partial", 1)       /* 0: q4 takes whole rounds only */                                          \
  X(OPT_Q4_DYNAMIC, "q4_dynamic", 0)       /* 1: per-XCD tile counters (the trainer sets it under data parallelism) */ \
  X(OPT_Q4_VAR, "q4_var", 0)               /* 1 / 2: alternative instruction schedules of the q4 K loop (same results) */ \
  X(OPT_GEMM_PP2, "gemm_pp2", 2)           /* 0: one workgroup per tile; 1: launches of two rounds or more only */     \
  X(OPT_PP2_DYNAMIC, "pp2_dynamic", 1)                                                                                  \
  X(OPT_PP2_GRID, "pp2_grid", 0)           /* resident workgroups per ping-pong launch (0: one per CU) */               \
  X(OPT_GEMM_PERSIST, "gemm_persist", 0)                                                                                \
  X(OPT_GEMM_TILE, "gemm_tile", 0)         /* force a tile shape of tell_gemm_nt (2, 5, 7, 8, 9; 0: the launcher's choice) */ \
  /* ---- small / grouped / K-major GEMMs, convolutions */                                                              \
  X(OPT_GEMM_S64, "gemm_s64", 1)           /* 0: never; 2: every K % 64 == 0 shape with M >= 512 */                     \
  X(OPT_GEMM_SMALL, "gemm_small", 1)                                                                                    \
  X(OPT_GEMM_RING, "gemm_ring", 0)         /* 2 / 3 / 4: LDS stages of the 64x64 direct-to-LDS tile everywhere */      \
  X(OPT_GEMM_SPLITK, "gemm_splitk", -1)                                                                                 \
  X(OPT_GROUP_TILE, "group_tile", 0)                                                                                    \
  X(OPT_GROUP_WIDE, "group_wide", 1)                                                                                    \
  X(OPT_GROUP_SORT, "group_sort", 1)                                                                                    \
  X(OPT_CONV_TILE, "conv_tile", 0)         /* 1 / 2 / 3 (+ 10: im2col path) forces a convolution tile shape */         \
  X(OPT_BN_FUSE, "bn_fuse", 1)                                                                                          \
  X(OPT_BN_COMBINE, "bn_combine", 0)                                                                                    \
  X(OPT_BN_WGS, "bn_wgs", 768)                                                                                          \
  /* ---- attention, DynamicConv, LayerNorm, softmax head, optimizer, generation step */                                \
  X(OPT_ATTN_TILE64, "attn_tile64", 0)                                                                                  \
  X(OPT_ATTN_SELF, "attn_self", 1)                                                                                      \
  X(OPT_ATTN_DMA, "attn_dma", 0)                                                                                        \
  X(OPT_ATTN_OCC, "attn_occ", 2)                                                                                        \
  X(OPT_DYNCONV_LDS, "dynconv_lds", 1)                                                                                  \
  X(OPT_DYNCONV_BLOCK, "dynconv_block", 1)                                                                              \
  X(OPT_LN_VAR, "ln_var", -1)              /* -1: the launcher's choice */                                              \
  X(OPT_ARGMAX_REGS, "argmax_regs", 1)                                                                                  \
  X(OPT_ADAM_VAR, "adam_var", -1)          /* -1: the built-in default shape */                                         \
  X(OPT_ADAM_GRID, "adam_grid", 4096)                                                                                   \
  X(OPT_SK_ROWS, "sk_rows", 0)             /* 128: 128-row skinny-linear workgroups above 64 rows (default: 32 rows everywhere) */ \
  X(OPT_SK_SPLIT, "sk_split", 0)           /* 1 / 2: share a long reduction of a skinny linear between 4 / 2 workgroups (split_ws) */ \
  X(OPT_SK_TALL_WAVES, "sk_tall_waves", 4) /* 8: the 128-row skinny workgroup as 8 waves x K / 8 */                   \
  X(OPT_SK_STAGED, "sk_staged", 1)         /* 0: the skinny linears load their MFMA fragments straight from memory */

// wrong-result timing probes: probe build only
#define TELL_PROBE_LIST(X)                                                                                              \
  X(PROBE_Q4_ABL, "q4_abl", 0)             /* 1 no epilogue, 2 no global stores, 3 s_memtime stamps into aux */        \
  X(PROBE_Q4E_VAR, "q4e_var", -1)          /* >= 0: the stamped statement */                                            \
  X(PROBE_PP2_ABL, "pp2_abl", 0)                                                                                        \
  X(PROBE_DCB_ABL, "dcb_abl", 0)                                                                                        \
  X(PROBE_SK_STAMP_PTR, "sk_stamp_ptr", 0) /* device address of the skinny linears' per-workgroup time stamps */       \
  X(PROBE_SK_STAMP_SLOTS, "sk_stamp_slots", 0)

enum TellOpt {
#define TELL_X(e, k, d) e,
  TELL_OPTION_LIST(TELL_X)
#ifdef TELL_PROBES
  TELL_PROBE_LIST(TELL_X)
#endif
#undef TELL_X
  TELL_OPT_COUNT
};

extern std::atomic<long> g_tell_opt[TELL_OPT_COUNT];      // api.hip
static inline long tell_opt(int o) { return g_tell_opt[o].load(std::memory_order_relaxed); }
#ifdef TELL_PROBES
#define tell_probe(e) tell_opt(e)
#else
#define tell_probe(e) 0L
#endif
