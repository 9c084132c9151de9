// persist.h — one denoiser call of the sampling loop (reference model.py:139-182 stroke path, heads and scheduler step:
// inference.py:84-94) as ONE persistent launch: the plan a launch reads, shared by persist.hip (device) and dhw_api.cpp (host).
//
// The per-kernel form runs the call as 11 launches; each boundary is a device-wide drain plus a cold start (the next
// kernel's workgroups fetch their input tiles and first weights with nothing to overlap: profiles/r04_encbc_wave_timeline_*.log,
// 2.2-2.8 us from a workgroup's first instruction to its first staged tile, on top of the 2.3-3 us a one-round grid lives
// longer than its workgroups).  Every dependency of the stroke path is INSIDE one sample (conv halos, self-attention over the
// sample's rows, skip connections), so the persistent form needs no device-wide barrier: a phase's tile waits for the
// previous phase's tiles of ITS sample only.
//
//   * one workgroup per CU (512 threads), resident for the whole call; phases = the per-kernel launches, in order;
//   * a sample belongs to ONE XCD for the whole call (samples [x * spx, (x + 1) * spx) to XCD x) and its tiles are dequeued
//     only by workgroups that read x from HW_REG_XCC_ID: producer and consumer of every hand-off share an L2 by
//     construction, whatever the dispatcher's placement, so the hand-off needs no L2 write-back (the agent-scope release
//     costs 1.7-6.5 us per block with dirty tiles: MI355X_MICROARCH.md) — only: producer stores -> every wave's
//     s_waitcnt vmcnt(0) -> workgroup barrier -> one agent-scope atomic add on the sample's counter; consumer: one lane
//     invalidates the CU's L1 (buffer_inv sc1), polls the counter (agent-scope relaxed loads), waits for the invalidate,
//     workgroup barrier, then plain loads.  Between the invalidate and those loads the workgroup touches no activation,
//     so its L1 cannot hold a stale line whenever the invalidate was issued.
//   * tiles are handed out by ONE ticket counter per XCD for the whole call (ticket order = phase order, so a tile's
//     dependencies always hold smaller tickets and the smallest unfinished ticket can always run: no deadlock for any
//     number of resident workgroups per XCD); a workgroup draws its next ticket while its stores drain, so the draw's
//     round trip (1.5-3 us under load) is off the critical path.  A producer that never arrives trips the consumer's
//     bounded spin (error word = 1 + phase).  An XCD that owns samples but got no workgroup of the launch (partitioned or
//     CU-masked device) makes nobody wait — its tiles are simply never drawn — so the LAST workgroup to leave compares every
//     XCD's ticket counter with its tile count and sets the error word (0x100 + XCD) when one fell short.  Either way the
//     launch finishes instead of hanging and the word (host-mapped memory) is read by the next dhw_sample call;
//   * the counters are zeroed by the last workgroup to leave, for the next launch (stream order makes that visible).
#pragma once
#include "dhw_kernels.h"

enum StepPhaseKind {
  PK_CONV_ENC1 = 0,   // convblock<126 rows, 128 <- 128>, input Linear fused (strokes)
  PK_CONV_ENC2A,      // convblock<62 rows, 192 <- 128> + enc3's first half
  PK_BC192,           // enc3 second half (64-row tiles)
  PK_CONV_ENC4,       // convblock<46 rows, 256 <- 192>
  PK_A256,            // enc5 first half (32-row tiles)
  PK_BC256_N2,        // enc5 second half + AvgPool + att_dense + first attention layer's first half
  PK_BC384_N1,        // attention layer second half + next layer's first half (16-row tiles)
  PK_BC384,           // last attention layer's second half
  PK_CONV_DEC3,       // decoder blocks with the fused Upsample + skip_conv input stage
  PK_CONV_DEC2,
  PK_CONV_DEC1,       // + eps / pen heads + scheduler step
  PK_COUNT
};
constexpr int STEP_MAX_PHASES = 16;
constexpr int STEP_XCDS = 8;

struct StepPhase {
  int kind;      // StepPhaseKind
  int tps;       // row tiles per sample
  int rows;      // rows per tile (tile r starts at row r * rows)
  int pad;
  ConvBlockParams cb;   // PK_CONV_*
  EncLayerParams el;    // PK_A* / PK_BC*
  EncChain nx;          // what the phase chains behind its own block (mode 0 = nothing)
};
struct StepPlan {
  int nphase, B;
  int spx;                     // samples per XCD: XCD x owns samples [x * spx, min(B, (x + 1) * spx))
  int pad;
  int cum_tps[STEP_MAX_PHASES + 1];   // tiles per sample of all earlier phases: ticket g of an XCD with ns samples is tile g - ns * cum_tps[ph] of phase ph
  unsigned* sync;              // step_sync_words(B) words, zero before the first launch: tickets, per-sample counters, exit count
  unsigned* err;               // host-mapped error word (0 = ok)
  unsigned long long* trace;   // diagnostics (DHW_PERSIST_TRACE builds): [workgroup][phase][4] s_memrealtime stamps, or null
  StepPhase ph[STEP_MAX_PHASES];
};
inline size_t step_sync_words(int B) { return (size_t)STEP_XCDS * 16 + (size_t)STEP_MAX_PHASES * B + 16; }   // tickets (one 64-byte line per XCD), counters, exit count

// tile geometry of a phase kind at stroke length L (the rows a tile of that kind covers; rows per sample at that level)
bool step_kind_geometry(int kind, int L, int* rows_per_tile, int* level_rows);
hipError_t persist_init();
// d_plan: device pointer to ONE StepPlan; grid: resident workgroups to start (<= the device's CU count)
hipError_t launch_step(const StepPlan* d_plan, int grid, hipStream_t st);
