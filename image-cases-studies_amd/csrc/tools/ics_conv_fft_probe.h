// ics_conv_fft_probe.h -- measurement hooks of ics_conv_fft.hip, for the harness builds of tools/bench_conv_fft.hip ONLY (never part of
// libics_hip.so: the library is built without ICS_FFT_PROBES and every hook below is then an empty macro defined in ics_conv_fft.hip).
//   -DICS_FFT_TRACE            per-wave phase timeline: shader-clock stamps at the stage boundaries of a unit (IcsFftArgs::trace)
//   -DICS_FFT_ABL_NOMEM=mask   a unit without some of its global memory traffic: 1 spectrum, 2 epilogue operands (mode 1: tile 0's), 4 window,
//                              8 stores, 16 mode 1's operands of tile 1 -- loads return a value made of their address, stores are dropped
//   -DICS_FFT_ABL_NOMATH       the stages' LDS traffic without their butterflies
//   -DICS_FFT_ABL_SKIP_BF      a unit without its two radix-8 column passes (what an LDS round trip with its barrier costs; results wrong)
//   -DICS_FFT_STAGGER=n        workgroups start n * 127 sleep ticks apart (measured: no effect)
// What was measured with them: NOTES_r05.md "Where a unit's time goes".
#pragma once
#ifndef ICS_FFT_ABL_NOMEM
#define ICS_FFT_ABL_NOMEM 0
#endif
#define ICS_FFT_PROBE_SKIP_LOAD(KIND, vi, si) if ((KIND) & ICS_FFT_ABL_NOMEM) { const float x_ = (float)((vi) + (si)); return (v4f){x_, x_, x_, x_}; }
#define ICS_FFT_PROBE_SKIP_STORE(v, vi, si) if (ICS_FFT_ABL_NOMEM & 8) { asm volatile("" :: "v"(v), "v"(vi), "s"(si)); return; }
#if defined(ICS_FFT_ABL_NOMATH) && defined(__HIP_DEVICE_COMPILE__)
#define ICS_FFT_PROBE_SKIP_MATH() return
#else
#define ICS_FFT_PROBE_SKIP_MATH() do { } while (0)
#endif
#ifdef ICS_FFT_ABL_SKIP_BF
#define ICS_FFT_PROBE_COLUMN_PASS(x) do { } while (0)
#else
#define ICS_FFT_PROBE_COLUMN_PASS(x) do { x } while (0)
#endif
#ifdef ICS_FFT_TRACE
#define ICS_FFT_PROBE_TRACE_DECL() int round_ = 0
#define ICS_FFT_STAMP(i) do { if ((tid & 63) == 0 && round_ < 16 && a.trace) a.trace[(((size_t)blockIdx.x * 16 + round_) * 16 + (tid >> 6)) * 10 + (i)] = (long long)__builtin_readcyclecounter(); } while (0)
#define ICS_FFT_PROBE_TRACE_NEXT() ++round_
#else
#define ICS_FFT_PROBE_TRACE_DECL() do { } while (0)
#define ICS_FFT_STAMP(i) do { } while (0)
#define ICS_FFT_PROBE_TRACE_NEXT() do { } while (0)
#endif
#ifdef ICS_FFT_STAGGER
#define ICS_FFT_PROBE_STAGGER() for (int i_ = 0; i_ < (int)((blockIdx.x >> 3) & 3) * ICS_FFT_STAGGER; ++i_) __builtin_amdgcn_s_sleep(127)
#else
#define ICS_FFT_PROBE_STAGGER() do { } while (0)
#endif
