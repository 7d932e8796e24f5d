/* mw_comms.h -- C-ABI exchange layer for a Fortran (or C) host: the collectives of the reference's
 * `module comms` (comms_mpi.f90) over RCCL, one process per GPU.  libmw_comms.so; the Fortran module that binds it
 * and provides the reference's own routine names is mc_water_ls_mw_amd/fortran/comms_rccl.f90.
 *
 * Replaces, routine by routine:
 *   mw_comms_init            Comms_Initialise               comms_mpi.f90:26-71    (mpi_init, comm_size, comm_rank)
 *   mw_comms_allreduce_sum   the MPI_Allreduce(MPI_SUM) of comms_allreduce_eta / _hist / _uhist
 *                                                            comms_mpi.f90:266,483,519
 *   mw_comms_allreduce3      the three of them in ONE collective (for a host that may fuse the calls of
 *                            mc_moves.F90:264-268; 2.4 KB at nbins = 101 is all latency)
 *   mw_comms_allreduce_max   comms_get_max                  comms_mpi.f90:279-297
 *   mw_comms_allgather       the Recv loop of comms_join_eta / comms_join_uhist (every rank receives every window and
 *                            stitches them itself, in rank order)            comms_mpi.f90:299-459
 *   mw_comms_bcast           Comms_BcastReal/Int/Log/Char   comms_mpi.f90:107-166,224-242
 *   mw_comms_sendrecv        comms_p2preal / comms_p2pint   comms_mpi.f90:168-222
 *   mw_comms_barrier         comms_barrier                  comms_mpi.f90:601-618
 *   mw_comms_finalize        Comms_Finalise                 comms_mpi.f90:569-599
 *   mw_comms_abort           (the failure path of every routine: the reference stops on an MPI error)
 *
 * Bootstrap (no MPI, no launcher dependency): rank and size come from the environment -- MW_COMMS_RANK /
 * MW_COMMS_SIZE, else RANK / WORLD_SIZE as torchrun sets them -- and the RCCL unique id travels through a file:
 * rank 0 creates it (ncclGetUniqueId makes no GPU call), writes MW_COMMS_ID_FILE (default
 * /tmp/mw_comms_id.<MASTER_PORT or 0>) under a temporary name and renames it into place; the others poll for it.
 * The GPU is MW_COMMS_DEVICE, else LOCAL_RANK, else rank modulo the visible devices.  Only then is the device
 * touched (hipSetDevice, stream, staging buffers, ncclCommInitRank).
 *
 * All buffers are HOST pointers (the Fortran host keeps its tables in host memory, 808 bytes each); they are staged
 * through one pinned host buffer and one device buffer on the library's own stream.  Every call is blocking and
 * collective: all ranks must make the same calls in the same order.  Return 0 = ok; otherwise mw_comms_last_error().
 * A size-1 job still goes through RCCL (N = 1 and N = 8 run the same code).
 */
#ifndef MW_COMMS_H
#define MW_COMMS_H

#ifdef __cplusplus
extern "C" {
#endif

int mw_comms_init(int *rank_out, int *size_out);
int mw_comms_allreduce_sum(double *buf, int n);
int mw_comms_allreduce3(double *a, double *b, double *c, int n);      /* c may be NULL (no unbiased histogram) */
int mw_comms_allreduce_max(double *buf, int n);
int mw_comms_allgather(const double *mine, double *all, int n);       /* all: size x n, rank-major */
int mw_comms_bcast(void *buf, long nbytes, int root);
int mw_comms_sendrecv(void *buf, long nbytes, int snode, int rnode);  /* snode sends buf, rnode receives into buf */
int mw_comms_barrier(void);
int mw_comms_finalize(void);
int mw_comms_abort(void);     /* failure path: drop the id file, ncclCommAbort, wait for nobody (the caller then stops) */
int mw_comms_rank(void);
int mw_comms_size(void);
const char *mw_comms_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
