// conv3_knobs.h - the scheduling / ablation knobs conv3.hip was tuned with (rounds 2-3), in ONE place.
//
// Product build (no MV2_EXPERIMENTS): every knob is the constant below and defining one on the command line is a compile error, so the
// product library has exactly one configuration of this parity-critical file.  Experiments build
// (magvit2_pytorch_amd/build.py --experiments, tools/build_c3_variants.sh): each knob may be overridden with -DMV2_C3_<KNOB>=<value>;
// what each one does and what it measured is recorded next to its default.
#pragma once

#ifndef MV2_EXPERIMENTS
#if defined(MV2_C3_RING) || defined(MV2_C3_HSPLIT) || defined(MV2_C3_PRIO) || defined(MV2_C3_ABL) || defined(MV2_C3_AGPR) || defined(MV2_C3_HPRO) || defined(MV2_C3_LATEPF) || defined(MV2_C3_OWNQ) || defined(MV2_C3_HTAB) || defined(MV2_C3_SKEW_DEFAULT) || defined(MV2_C3_PSETS) || defined(MV2_C3_HWT)
#error "MV2_C3_* knobs are experiment switches: build with -DMV2_EXPERIMENTS (magvit2_pytorch_amd/build.py --experiments) to change them"
#endif
#endif

// Weight fragments (1 KB per wave, 16 B per lane) stream global -> VGPR through a ring of RING slots: a group consumes 18
// fragments (9 taps x 2 channel tiles), fragment f lives in slot f % RING and its slot is refilled with fragment f + RING right
// behind its last MFMA, so RING must divide 18 for the slot of a fragment to be a compile-time constant.  RING = 6 is a
// prefetch distance of 3 taps, RING = 9 of 4.5 taps (12 more registers; paid for by HSPLIT).
#ifndef MV2_C3_RING
#define MV2_C3_RING 6
#endif
// HSPLIT: the next group's halo travels in two halves (rows 0-2 issued at tap 0 and written at tap 4, rows 3-5 issued at tap 4
// and written at tap 8) instead of six loads parked in registers from tap 0 to tap 5: 12 registers less inside the tap loop.
#ifndef MV2_C3_HSPLIT
#define MV2_C3_HSPLIT 0
#endif
// PRIO (experiments): the two workgroups of a CU start together and the older one wins every issue arbitration, so it runs at its
// solo rate (~400 clocks per tap), finishes early, and the younger one does most of its taps alone afterwards (k3_bench residency
// report: 80k against 122k clocks for equal work).  PRIO = 1 alternates s_setprio between the pair per 9-tap group, PRIO = 2 per tap
// (both measured: the pair ends closer, 96k / 119k, the launch is not faster — a group counter is not a shared clock, whoever is
// ahead by an odd number of groups has the partner's priority and wins on age).  PRIO = 3 (not yet measured) slices by the CU's
// own clock: bit 13 of s_memtime picks which workgroup of the pair is favoured, whatever their progress.
#ifndef MV2_C3_PRIO
#define MV2_C3_PRIO 0
#endif
// ABL (experiments, TIMING ONLY — results are wrong): 1 = the a = 1 weight fragments are never reloaded (-37 % of the bytes a wave
// pulls through L1), 2 = no weight fragment is reloaded (-75 %), 3 = the position fragments are not re-read from LDS, 4 = 2 + 3.
// Separates "bound by vector-memory throughput" from "bound by issue / dependencies" in the tap loop.
#ifndef MV2_C3_ABL
#define MV2_C3_ABL 0
#endif
// AGPR (experiments): the tap loop's MFMAs are emitted through inline asm with the accumulator constrained to the accumulator
// register file ("+a").  Under the 256-register budget the compiler selects the VGPR form of the MFMA (C / D in arch VGPRs,
// .agpr_count 0); the measured ~49-cycle solo cadence of those MFMAs against the pipe's 32 may be the price of that form
// (tools/mfma_cadence.cpp measures it in isolation).  bf16 only; the accumulators are copied to VGPRs for the reduce.
#ifndef MV2_C3_AGPR
#define MV2_C3_AGPR 0
#endif
// PSETS (experiments): register sets of position fragments.  3 (product) lets set (TP + 2) % 3 rest for a tap; 2 is a plain double
// buffer (16 registers less): 9 taps per group is odd, so the fragments of the next group's first tap are moved from set 1 to
// set 0 at the end of tap 8 (16 v_mov per group).
// HPRO (experiments): the first halo of a segment travels in its own registers and the in-loop halo code re-derives its lane
// values from an opaque copy of the thread id.  Needed by the AGPR form: with 128 arch VGPRs the allocator otherwise spills the
// in-loop halo registers (scratch stores / reloads with vmcnt(0) inside the tap loop); with HPRO the tap loop has no scratch access.
#ifndef MV2_C3_HPRO
#define MV2_C3_HPRO 0
#endif
// LATEPF (experiments, for the AGPR form): where the next segment's first halo / weights / bias are requested.  0 (product): behind the
// cross-wave reduce, so that they land under the ticket and the epilogue — 52 registers live across those phases, which the 128 arch
// VGPRs of the AGPR form cannot hold (the loaded halo is spilled to scratch and reloaded: ~100 spill slots).  1: after the epilogue.
// 2: at the top of the next segment, nothing loop-carried in registers (27 spill slots left, none inside the tap loop).
#ifndef MV2_C3_LATEPF
#define MV2_C3_LATEPF 0
#endif
#ifndef MV2_C3_OWNQ
#define MV2_C3_OWNQ 0
#endif
// HTAB: the six halo row offsets of a thread are kept in a thread-private LDS table, written once per SEGMENT (tile), instead of being
// re-derived per GROUP.  The per-group derivation is, per row, a division by 18 (v_mul_hi_i32), two 64-bit multiply-adds and an exec
// branch: ~70 issue cycles per row = ~450 per group at the top of tap 0, where a single wave per SIMD has no MFMA in flight to hide
// them (the ISA of the round-2 kernel; a group is 9 x 256 MFMA cycles).  With the table a row costs one saturating add: entries of rows
// in the spatial zero padding hold 0xffffffff, which stays out of range under uadd.sat whatever the group's frame / chunk offset.
// 1: table reads at the top of tap 0, the six loads between the two MFMA batches of tap 0.  2: loads spread over taps 0-2 (two rows
// each), LDS writes over taps 5-7.  The table lives behind the two halo images inside the reduce region (free while taps run).
#ifndef MV2_C3_HTAB
#define MV2_C3_HTAB 1      // measured r03g (tools/k3_bench levels): 84.1 / 77.7 / 72.0 / 71.5 / 41.5 / 26.5 -> 83.4 / 75.5 / 71.3 / 70.0 / 40.3 / 24.7 us
#endif
#ifndef MV2_C3_SKEW_DEFAULT
#define MV2_C3_SKEW_DEFAULT 0
#endif
#ifndef MV2_C3_PSETS
#define MV2_C3_PSETS 3
#endif
#ifndef MV2_C3_HWT
#define MV2_C3_HWT 5      // tap at which the prefetched halo goes into the idle image (HSPLIT = 0); <= 7: tap 8 opens with the barrier
#endif
