/* openmm_hip_comm.h -- the collective layer of the multi-GPU ("one 1M-atom box on N MI355X") path of the HIP
 * platform, part of the same thin C ABI as openmm_hip_kernels.h (implemented in libopenmm_hip_kernels.so).
 *
 * One process drives one GPU; the N processes of a run form one communicator.  Two transports:
 *
 *   RCCL      the product transport: ncclAllGather / grouped ncclSend+ncclRecv over xGMI, enqueued on the caller's
 *             HIP stream (stream ordered, no host synchronisation).  librccl.so.1 is opened with dlopen when the first
 *             communicator is created, so single-GPU users never load it.  The launcher (bench.py, a user's mpirun
 *             wrapper) creates the 128-byte ncclUniqueId on rank 0 with ommhip_comm_unique_id() and hands it to every
 *             rank, e.g. through torch.distributed's store; the plugin receives it as the platform property "CommId".
 *   callback  test transport: collectives are staged through host memory and performed by a function the caller
 *             supplies (tests: torch.distributed with the gloo backend).  It lets the whole decomposed step run with
 *             world_size 2 on the CPU emulator and on a single GPU shared by two processes, where RCCL refuses to form
 *             a communicator ("Duplicate GPU detected").
 *
 * What the reference does instead: platforms/cuda/src/CudaParallelKernels.cpp:143-254 drives all devices from one
 * process with per-device worker threads and copies positions / forces through device 0 with peer memcpys
 * (CudaParallelKernels.cpp:178-202); there is no collective library on its path.
 *
 * All functions return 0 on success, a hipError_t, or 1000 + ncclResult_t for RCCL failures (ommhip_error_string
 * understands all three).
 */
#ifndef OPENMM_HIP_COMM_H_
#define OPENMM_HIP_COMM_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ommhip_comm ommhip_comm;

/* Host all-gather supplied by the caller of the callback transport: every rank passes `bytes` bytes in `send` and
 * receives size * bytes in `recv`, rank-major.  Must return 0 on success. */
typedef int (*ommhip_host_all_gather_fn)(void* user, const void* send, void* recv, size_t bytes);

#define OMMHIP_COMM_ID_HEX_LEN 257      /* 128 bytes as hex + NUL */

/* rank 0: a fresh ncclUniqueId as a hex string (hex must hold OMMHIP_COMM_ID_HEX_LEN chars) */
int ommhip_comm_unique_id(char* hex);
/* collective over all ranks (blocks until every rank has called it); the current HIP device is the rank's GPU */
int ommhip_comm_create_rccl(const char* id_hex, int rank, int size, ommhip_comm** comm);
int ommhip_comm_create_callback(ommhip_host_all_gather_fn fn, void* user, int rank, int size, ommhip_comm** comm);
/* Diagnostics ("alone" transport, bench.py --rank-alone): rank `rank` of `size` exists, its peers do not -- every collective returns at once and
 * moves nothing (the halo keeps the positions it was given at the start).  What the rank computes is meaningless; what it LAUNCHES is one rank's
 * step of the decomposition with its two streams overlapped as over RCCL and no communication at all: the compute side of the scaling limit. */
int ommhip_comm_create_alone(int rank, int size, ommhip_comm** comm);
/* A second communicator over the same ranks (collective): for traffic issued from another stream (reciprocal space runs
 * beside the pair kernel).  RCCL: ncclCommSplit with one colour; the callback transport shares its callback. */
int ommhip_comm_duplicate(ommhip_comm* comm, ommhip_comm** copy);
int ommhip_comm_destroy(ommhip_comm* comm);
int ommhip_comm_rank(const ommhip_comm* comm);
int ommhip_comm_size(const ommhip_comm* comm);
const char* ommhip_comm_transport(const ommhip_comm* comm);      /* "rccl" or "callback" */

/* In-place all-gather of device memory: buffer_d holds size * bytes_per_rank bytes, this rank's part already sits at
 * offset rank * bytes_per_rank.  (Positions after the integration step, forces/velocities when a State is downloaded.) */
int ommhip_comm_all_gather(ommhip_comm* comm, void* buffer_d, size_t bytes_per_rank, void* stream);
/* All-to-all of device memory: chunk p (bytes_per_pair bytes) of send_d goes to rank p and arrives as chunk `rank` of its
 * recv_d.  (The two transposes of the slab-decomposed 3-D FFT.) */
int ommhip_comm_all_to_all(ommhip_comm* comm, const void* send_d, void* recv_d, size_t bytes_per_pair, void* stream);
/* Ring neighbours (periodic slabs): send bytes_down to rank-1 and bytes_up to rank+1; receive bytes_down from rank+1
 * (what it sent down) into recv_from_up_d and bytes_up from rank-1 into recv_from_down_d.  (Halo planes of the PME
 * potential.)  With size == 1 the data is copied locally. */
int ommhip_comm_ring_exchange(ommhip_comm* comm, const void* send_down_d, void* recv_from_up_d, size_t bytes_down,
                              const void* send_up_d, void* recv_from_down_d, size_t bytes_up, void* stream);
/* Halo exchange between neighbouring slabs (periodic ring) -- the per-step position exchange of a decomposed run.  buffer_d holds
 * one range of `rank_stride` bytes per rank (slot order: rank r's records start at r * rank_stride); the plan -- identical on all
 * ranks -- says, for EVERY rank, which part of its range its lower neighbour needs ("down" section) and which part its upper
 * neighbour needs ("up" section; the two may overlap).  In place: this rank's own range is the send buffer, the sections of rank - 1
 * and rank + 1 land at their natural addresses inside this rank's copy of the buffer.  `trailer_bytes` > 0: the small record at
 * trailer_offset of every rank's range (momentum, flags) goes to ALL ranks in the same group (size - 1 tiny sends).
 * RCCL: one ncclGroup of at most 2 + (size - 1) sends and as many receives -- every one over a link of its own on xGMI.
 * With two ranks both neighbours are the same peer; with one rank nothing moves. */
#define OMMHIP_MAX_RANKS 64
typedef struct ommhip_halo_plan {
    size_t rank_stride;
    size_t down_offset[OMMHIP_MAX_RANKS], down_bytes[OMMHIP_MAX_RANKS];     /* relative to the start of the rank's range */
    size_t up_offset[OMMHIP_MAX_RANKS], up_bytes[OMMHIP_MAX_RANKS];
    size_t trailer_offset, trailer_bytes;
} ommhip_halo_plan;
int ommhip_comm_halo_exchange(ommhip_comm* comm, void* buffer_d, const ommhip_halo_plan* plan, void* stream);
/* The way back: what a rank accumulated on atoms of its LOWER neighbour goes home.  With half-shell evaluation a pair of
 * atoms of two neighbouring slabs is evaluated once, by the upper rank, which holds the lower rank's "up" section as part of its halo: the
 * force on those atoms sits in the upper rank's fixed-point buffer and is returned here.  force_d: the platform's SoA force buffer, three
 * components of padded_slots 64-bit elements in slot order.  The plan (identical on all ranks) names the up section of EVERY rank as
 * global slot indices.  This rank sends, per component, the elements of its lower neighbour's section to that neighbour and receives
 * those of its own section from its upper neighbour into staging_d (3 x num_slots[rank] elements, component-major), which a kernel then
 * adds to force_d -- integer addition, so the sum does not depend on arrival order.  RCCL: one group of three sends and three receives.
 * With one rank nothing moves. */
typedef struct ommhip_halo_return_plan {
    int first_slot[OMMHIP_MAX_RANKS], num_slots[OMMHIP_MAX_RANKS];
} ommhip_halo_return_plan;
int ommhip_comm_halo_return(ommhip_comm* comm, long long* force_d, int padded_slots, const ommhip_halo_return_plan* plan, long long* staging_d, void* stream);
/* Host all-gather of small records (energies, momenta, flags): blocking; every rank then reduces the size records in the
 * same order, which makes sums bit-identical on all ranks. */
int ommhip_comm_all_gather_host(ommhip_comm* comm, const void* send, void* recv, size_t bytes, void* stream);

/* Diagnostics (callback transport, environment OMMHIP_COMM_DIAG=1): seconds spent inside collectives since the last reset,
 * each counted from the moment the device is idle.  Used by bench.py --serialize-ranks to time a rank's step without
 * communication on a box with fewer GPUs than ranks. */
double ommhip_comm_diag_seconds(int reset);

#ifdef __cplusplus
}
#endif
#endif
