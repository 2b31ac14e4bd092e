// gett_grouped.hip -- the grouped launches of gett.hip (the (T) GEMMs: 85 % of a config-5 step) in a translation unit of their
// own, so that they can be compiled with a scheduling strategy of their own: hipcc's `max-memory-clause` strategy
// (-mllvm -amdgpu-sched-strategy=max-memory-clause, set in the Makefile for this file only) makes the grouped 256 x 128 kernel
// 2.0 % faster ((T) 506 -> 497 ms at config 5, three alternating runs per build in one GPU session) and the plain tiles 1 % slower
// (ring 62.0 -> 61.2 TF), so gett.hip keeps the default.  Everything but gett_launch_grouped / preload_gett_grouped is compiled
// out of this unit (AFESP_GETT_GROUPED_TU).
#define AFESP_GETT_GROUPED_TU 1
#include "gett.hip"
