// ics_synth_gradk_probe.h -- measurement hooks of ics_synth_gradk_mfma.hip, for the harness builds of tools/bench_synth_gradk.hip ONLY (never
// part of libics_hip.so: the library is built without ICS_FUSED_PROBES and every hook is then an empty macro / a constant 0).
//   -DICS_FUSED_TIMING       per-wave cycle totals between the marks of a tile
//   -DICS_FUSED_TRACE        per-wave phase timeline (format of ics_conv_mfma.hip's ICS_MFMA_TRACE, scripts/trace_conv_mfma.py)
//   -DICS_FUSED_ABLATE=mask  a tile without some of its work: 1 no gradient loop, 2 no convolution loop, 4 no e' planes, 8 no conversion of
//                            channels 1, 2, 16 no image operand, 32 no funnel shifts (operands taken unshifted), 64 no MFMAs (operands kept
//                            alive), 128 no workgroup barriers
// What was measured with them: NOTES_r03.md 4c, NOTES_r04.md 4d.
#pragma once
// phase timing probe (tools/bench_synth_gradk.hip -DICS_FUSED_TIMING): per-wave cycle totals between the marks
#ifdef ICS_FUSED_TIMING
__device__ unsigned long long ics_fused_ticks[17];
#define FTICK_INIT unsigned long long tk_prev = __builtin_readcyclecounter(), tk_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}
#define FTICK(i) do { const unsigned long long tk_now = __builtin_readcyclecounter(); tk_acc[i] += tk_now - tk_prev; tk_prev = tk_now; } while (0)
#define FTICK_FLUSH do { if ((threadIdx.x & 63) == 0) { for (int i = 0; i < 16; ++i) atomicAdd(&ics_fused_ticks[i], tk_acc[i]); atomicAdd(&ics_fused_ticks[16], 1ull); } } while (0)
#elif defined(ICS_FUSED_TRACE)
// phase timeline (tools/bench_synth_gradk.hip -DICS_FUSED_TRACE; format of ics_conv_mfma.hip's ICS_MFMA_TRACE, scripts/trace_conv_mfma.py):
// lane 0 of every wave records (100 MHz wall clock << 8 | mark) at each mark; entry 0 = HW_ID | XCC_ID << 32.  1024 entries per wave.
__device__ unsigned long long* ics_fused_trace_buf;
#define FTICK_INIT unsigned long long* tr_ = ics_fused_trace_buf + ((size_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * 1024; int tri_ = 0; \
  if ((threadIdx.x & 63) == 0) { tr_[tri_++] = (unsigned long long)__builtin_amdgcn_s_getreg(63492) | ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32); tr_[tri_++] = (wall_clock64() << 8) | 15; }
#define FTICK(i) do { if ((threadIdx.x & 63) == 0 && tri_ < 1023) tr_[tri_++] = (wall_clock64() << 8) | (i); } while (0)
#define FTICK_FLUSH do { if ((threadIdx.x & 63) == 0) tr_[tri_] = 0; } while (0)
#else
#define FTICK_INIT
#define FTICK(i)
#define FTICK_FLUSH
#endif
#ifndef ICS_FUSED_ABLATE
#define ICS_FUSED_ABLATE 0
#endif
#define ICS_FUSED_ABL(mask) (ICS_FUSED_ABLATE & (mask))
