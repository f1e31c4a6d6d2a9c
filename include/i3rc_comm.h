/*
 * include/i3rc_comm.h -- process layer of the MI355X integrator: one process per GPU, sums over processes with RCCL.
 *
 * Stands where the reference has Code/multipleProcesses_mpi.f95 (MPI_INIT / COMM_RANK / COMM_SIZE :26-39, MPI_Barrier
 * :41-49, MPI_FINALIZE :51-55, MPI_REDUCE(MPI_REAL, MPI_SUM) :57-131).  The Fortran module MultipleProcesses of the
 * shell binds these entry points; see INTEGRATION.md.
 *
 * Ranks are taken from the environment of the usual launchers: RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR /
 * MASTER_PORT (torchrun style), else OMPI_COMM_WORLD_RANK / _SIZE / _LOCAL_RANK, else a single process.
 * Backends: "rccl" (default: ncclAllReduce over xGMI) and "shm" (I3RC_COMM_BACKEND=shm: POSIX shared memory on one
 * node, no GPU needed -- used by the CPU tests of the N > 1 path).  Bootstrap of both: rank 0 listens on
 * port MASTER_PORT + 1 (I3RC_COMM_PORT overrides it; MASTER_PORT itself belongs to the launcher's store under torchrun)
 * of MASTER_ADDR (an IPv4 address or a host name) and hands the ncclUniqueId / the segment name to the other ranks
 * behind a magic / size header, so that a connection to some other service fails instead of hanging; nothing is left
 * behind for a later run.
 * All functions return 0 on success; i3rc_comm_last_error() describes the last failure.
 */
#ifndef I3RC_COMM_H
#define I3RC_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int i3rc_comm_init(int *numProcs, int *thisProc);   /* initializeProcesses(numProcs, thisProcNum) */
int i3rc_comm_local_device(void);                    /* HIP device of this process (LOCAL_RANK), 0 for one process */
int i3rc_comm_barrier(void);                         /* synchronizeProcesses */
int i3rc_comm_sum_float(float *values, int64_t n);   /* sumAcrossProcesses: in place, result on every rank */
/* ... and in float64 (ncclDouble): no counterpart in the reference, whose MPI_REDUCE is MPI_REAL -- for the build's own driver, which
 * gathers a loop's statistics on the device in float64 (i3rc_hip_run_batches_moments) and sums them over the ranks in ONE packed
 * buffer without rounding them to real(4) first (fortran/tools/i3rcDriver.f95, sumAcrossProcesses for real(8) arrays). */
int i3rc_comm_sum_double(double *values, int64_t n);
int i3rc_comm_finalize(void);                        /* finalizeProcesses */
const char *i3rc_comm_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
