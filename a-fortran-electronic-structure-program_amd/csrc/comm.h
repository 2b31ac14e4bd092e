// comm.h -- the one collective of the path: a sum over the ranks (one process per GPU, one context each).
//
// The reference has no distributed layer; its (T) loop ends in an OpenMP `reduction(+: ...)` over threads
// (src/ccsd.f90:2091).  Here the (i<=j<=k) triples -- and, for world > 1, the column panels of the CCSD iteration's large
// products -- are split over ranks and that reduction becomes an all-reduce:
//   RCCL   ncclAllReduce(sum, fp64) on the context's stream, in place on device memory (xGMI between the GPUs of a node).
//          librccl.so.1 is opened on first use, so a single-GPU run has no dependency on it.
//   HOST   ranks of one node add through a file-backed shared segment (fixed rank order: every rank gets bit-identical
//          sums).  It exists to rehearse the rank logic where ranks share one GPU -- RCCL refuses two ranks on one
//          device -- and moves host memory only; device buffers make a round trip through pinned memory.
#pragma once
#include <stdint.h>

#include <string>

#include "afesp_internal.h"

namespace afesp {

struct Comm {
    int rank = 0, world = 1, transport = 0;
    // RCCL
    void* nccl_comm = nullptr;
    // HOST
    void* seg = nullptr;          // mmap'ed segment
    size_t seg_bytes = 0;
    int64_t slot_doubles = 0;     // capacity of one rank's slot
    int fd = -1;
    std::string path;
    uint32_t epoch = 0;           // barrier generation of this rank
    double* pinned = nullptr;     // staging for device buffers (HOST) and for host scalars (RCCL)
    double* dev_small = nullptr;  // device staging for host scalars (RCCL)
    int64_t pinned_doubles = 0;
};

void comm_unique_id(char id[128]);
Comm* comm_create(Context& cx, int rank, int world, int transport, const char* bootstrap_path, const char* unique_id);
void comm_destroy(Comm* c);
// in-place sums; every rank must call with the same n
void comm_allreduce_host(Context& cx, Comm* c, double* host, int64_t n);
void comm_allreduce_dev(Context& cx, Comm* c, double* dev, int64_t n);   // ordered on cx.stream
void comm_barrier(Context& cx, Comm* c);

}  // namespace afesp
