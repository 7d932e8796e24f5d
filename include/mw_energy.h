/* mw_energy.h -- C ABI of libmw_hip.so, the MI355X (gfx950) mW energy engine.
 *
 * This is the drop-in boundary for the hot path of keb721/mc_water_ls_mw:
 * the Fortran `module energy` (molint.F90:10-37).  Every entry point cites the
 * reference routine it stands behind; the replacement Fortran module
 * (mc_water_ls_mw_amd/fortran/energy_hip.F90) binds these with ISO_C_BINDING
 * and keeps the reference's public names, so mc_moves.F90 / main.f90 compile
 * against it unchanged.
 *
 * Conventions
 *   - plain C types only; no torch / HIP types cross this boundary
 *   - every function returns 0 on success, nonzero on failure; the message is
 *     then available from mw_last_error().  There is NO CPU fallback: without
 *     a usable gfx950 device mw_init fails.
 *   - indices are 1-based exactly as the Fortran host passes them: box
 *     (lattice) `ils` in 1..nboxes, molecule `imol` in 1..nwater, image 1 = the
 *     central cell.
 *   - lengths in bohr, energies in Hartree, double precision throughout.
 *   - the host owns ljr / hmatrix / model_energy; inputs are
 *     copied during the call, no pointer handed out outlives mw_finalize().
 *   - a "box" is one lattice of one walker.  The reference has 1 or 2 per
 *     process (userparams::num_lattices); the engine accepts any number so that
 *     many independent walkers share a launch (the *_batch / *_launch entries).
 *   - calls are synchronous unless named *_launch; one host thread per context.
 */
#ifndef MW_ENERGY_H
#define MW_ENERGY_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MW_MAXNEIGH_LIMIT 64   /* the per-atom in-range mask is one 64-bit word        */
#define MW_MAX_IVECT      1024 /* 10 bits of a packed list entry hold the image number */

/* ---- lifetime: energy_init / energy_deinit (molint.F90:91-153, 155-171) ---------- */

/* Allocate device state for `nboxes` boxes of `nwater` molecules each with
 * `maxneigh` list slots per molecule (reference: 50, molint.F90:79).  `device`
 * is the HIP device ordinal (the local rank in a one-process-per-GPU farm); -1
 * takes it from MW_DEVICE / LOCAL_RANK / OMPI_COMM_WORLD_LOCAL_RANK /
 * MV2_COMM_WORLD_LOCAL_RANK / MPI_LOCALRANKID / SLURM_LOCALID, modulo the
 * number of devices (0 if none is set). */
int mw_init(int device, int nwater, int nboxes, int maxneigh);
int mw_finalize(void);
int mw_is_initialised(void);

/* Message of the last failing call on this thread ("" if none). */
const char *mw_last_error(void);

/* Model constants as the engine holds them (molint.F90:63-74):
 * out = sigma, epsilon, lambda, A, B, gamma, a, cos0. */
int mw_constants(double out[8]);

/* ---- implicit inputs of the Fortran module: model::hmatrix, model::ljr ----------- */

/* compute_ivects(ils) (molint.F90:174-217): take hmatrix(:,:,ils) (column-major,
 * 9 doubles, column k = cell vector k), rebuild the image vectors and mirror
 * them on the device.  *nivect_out (may be NULL) receives nivect(ils). */
int mw_set_cell(int ils, const double h[9], int *nivect_out);
/* The same for `count` consecutive boxes in one call: h = count x 9 doubles, nivect_out (may be NULL) = count ints.
 * One host-to-device transfer per mirrored array for the whole range (energy_init's loop over lattices,
 * molint.F90:122-136, for a farm of thousands of boxes). */
int mw_set_cells_range(int first_ils, int count, const double *h, int *nivect_out);
/* Copy out ivect(:,1:nivect,ils) (3 doubles each); returns nivect via *nivect_out. */
int mw_get_ivects(int ils, double *out, int max_vectors, int *nivect_out);

/* Mirror all positions of box ils: xyz = &ljr(1,1,1,ils), nwater x 3 AoS. */
int mw_upload_positions(int ils, const double *xyz);
int mw_download_positions(int ils, double *xyz);
/* The same for `count` consecutive boxes in one transfer (count x nwater x 3). */
int mw_upload_positions_range(int first_ils, int count, const double *xyz);
int mw_download_positions_range(int first_ils, int count, double *xyz);
/* Mirror one molecule (trial move, silent revert: mc_moves.F90:1079,1186). */
int mw_patch_position(int ils, int imol, const double r[3]);

/* ---- compute_neighbours(ils) (molint.F90:501-559) -------------------------------- */

/* Rebuild the Verlet list of box ils from the mirrored positions and image
 * vectors: every (jmol, image) with |r_j + ivect - r_i|^2 < (1.18 rc)^2 except
 * (j = i, central image), ordered j ascending then image ascending -- the same
 * set in the same order as the reference.  Fails (nonzero) if any molecule has
 * more than maxneigh entries (the reference overflows silently).
 * min_nn / max_nn may be NULL. */
int mw_build_neighbours(int ils, int *min_nn, int *max_nn);
int mw_build_neighbours_batch(int first_ils, int count, int *min_nn, int *max_nn);
/* Copy the list out in the reference's layout: nn(1:nwater), jn/vn(1:maxneigh,
 * 1:nwater) (slot fastest), 1-based, unused slots 0.  Any pointer may be NULL. */
int mw_get_neighbours(int ils, int *nn, int *jn, int *vn);

/* ---- compute_model_energy(ils) (molint.F90:407-499) ------------------------------ */

int mw_model_energy(int ils, double *e);
/* compute_model_energy(ils) exactly as the host issues it: xyz = &ljr(1,1,1,ils) is mirrored first, then evaluated -- one
 * call, one synchronisation (mw_upload_positions + mw_model_energy cost two round trips; the host's volume move pays this
 * once per lattice per attempt, mc_moves.F90:1340). */
int mw_model_energy_of(int ils, const double *xyz, double *e);
/* Same for boxes first_ils .. first_ils+count-1 in one launch. */
int mw_model_energy_batch(int first_ils, int count, double *e_out);
/* In-range interaction counts of the last model-energy evaluation of box ils
 * as the reference enumerates them: directed pairs (molint.F90:454) and
 * i-centred triplets (molint.F90:477). */
int mw_model_energy_counts(int ils, long long *npairs, long long *ntriplets);
/* The same, summed over boxes first_ils .. first_ils+count-1. */
int mw_model_energy_counts_total(int first_ils, int count, long long *npairs, long long *ntriplets);
/* Sum of nn(imol) over those boxes: the number of list entries a full-box pass reads. */
int mw_neighbour_total(int first_ils, int count, long long *total_entries);

/* ---- compute_local_real_energy(imol, ils) (molint.F90:220-404) ------------------- */

/* Local energy of molecule imol from the mirrored positions. */
int mw_local_energy(int ils, int imol, double *e);
/* The drop-in form: first mirror up to two host positions (the molecule being
 * queried and the one queried just before it, whose host position may have
 * been silently reverted), then evaluate -- one launch.  imol_prev <= 0 or
 * r_prev == NULL skips the second patch. */
int mw_local_energy_patched(int ils, int imol, const double r_imol[3],
                            int imol_prev, const double r_prev[3], double *e);
/* The same call split in two, for a host that knows its NEXT question while it still waits for the answer to this
 * one -- the two lattices of a move, mc_moves.F90:1006-1018 and 1076-1092: `post` sends the request of lattice ils to
 * that lattice's mail slot and returns, `collect` waits for its reply.  One posted request per lattice at a time; any
 * other single call on that lattice waits for it first.  Both return 2 -- not an error, no message -- when there is
 * nothing to gain or to collect: the resident server is off (MW_LOCAL_SERVER=0), nothing was posted, or an entry
 * point that changes device state ran in between (the reply may predate it: ask again with mw_local_energy_patched). */
int mw_local_energy_post(int ils, int imol, const double r_imol[3], int imol_prev, const double r_prev[3]);
int mw_local_energy_collect(int ils, double *e);

/* Batched single-move path: request m is (ils[m], imol[m]) and, if trial_xyz
 * is not NULL, molecule imol[m] is evaluated AT trial_xyz[3m..3m+2] without
 * changing the mirrored positions (what the old/new pair of
 * mc_water_translation needs, mc_moves.F90:1010,1083). */
int mw_local_energy_batch(int n, const int *ils, const int *imol,
                          const double *trial_xyz, double *e_out);
/* e_old[m] with the mirrored position, e_new[m] at the trial position. */
int mw_delta_energy_batch(int n, const int *ils, const int *imol,
                          const double *trial_xyz, double *e_old, double *e_new);

/* ---- device-resident batch pipeline (what bench.py times) ------------------------ */
/* Requests are staged once; launches are asynchronous on the engine's stream;
 * results stay on the device until fetched. */
int mw_moves_upload(int n, const int *ils, const int *imol, const double *trial_xyz);
int mw_moves_launch(void);                       /* old + new local energies of every staged move */
int mw_moves_fetch(double *e_old, double *e_new);
/* Totals over the staged moves of the last launch: out = {interactions_old, slots_old,
 * interactions_new, slots_new}.  An interaction is an in-range pair or an in-range triplet
 * slot with cos(theta) < 0.99 (molint.F90:276,361,367); a slot is one list entry visited
 * (nn(i) + sum of nn(j) over in-range j), which prices the call's algorithmic bytes. */
int mw_moves_counts(long long out[4]);
int mw_model_energy_launch(int first_ils, int count);
/* Both launches of a step in one host call: the full-box energies of boxes first_ils .. first_ils+count-1, then the staged
 * moves (mw_model_energy_launch + mw_moves_launch).  timer_slot >= 0 brackets them with the engine's event timers:
 * slot timer_slot around the full-box kernel, timer_slot + 1 around the move kernels (read with mw_timer_elapsed_ms).
 * A step whose kernels take ~1.4 ms is otherwise preceded by six host calls, and a host that waits for an exchange
 * between steps (as the walker farm does) cannot issue them ahead. */
int mw_step_launch(int first_ils, int count, int timer_slot);
int mw_model_energy_fetch(int first_ils, int count, double *e_out);
int mw_build_neighbours_launch(int first_ils, int count);
int mw_sync(void);

/* ---- device-resident translation-move driver (mc_water_translation, mc_moves.F90:966-1213) ------
 * A walker is `nlat` (1 or 2) consecutive boxes: walker w owns boxes (w-1)*nlat+1 .. w*nlat.  Each
 * walker runs its own Markov chain on the device, one wavefront per walker: pick a molecule, trial
 * displacement in the active lattice mapped through fractional coordinates into the partner lattice
 * (mc_moves.F90:1042-1066), old/new local energy in every lattice, update of the order parameter mu,
 * multicanonical weights through eta_weight (mc_moves.F90:893-964, bins from mu_to_bin :2187-2215),
 * Metropolis acceptance, revert on rejection.  The per-box energies (the values mw_model_energy* left,
 * or mw_set_model_energy) are edited the way the reference's caller edits model_energy.  Lists are
 * NOT rebuilt inside: call mw_build_neighbours_batch at the host's list_update_int, as mc_cycle does.
 * Random numbers are Philox4x32-10 with counter (move index, walker, call) and key `seed`: move
 * `move0 + m` of walker w draws the same six numbers whatever the launch geometry. */
int mw_sweep_configure(int nlat, double beta, double max_trans_bohr, int nbins, int eta_interp,
                       int start_bin, int end_bin, double r_pos, double a_pos, double r_neg, double a_neg,
                       double mu_lo, double mu_hi,
                       const double *weight, const double *mu_bin, const double *binwidth);
/* The rest of a translation-only mc_cycle, per walker and on the device (two lattices): after every move
 * mc_update_wl_bins (mc_moves.F90:1597-1689, default schedule) on the walker's own histogram /
 * unbiased_hist / weight tables when `record` (mc_cycle_num >= eq_mc_cycles) -- with `samplerun` the
 * weights stay fixed and the unbiased histogram accumulates, otherwise the visited bin's weight grows by
 * av_binwidth*wl_factor/binwidth and the minimum is subtracted -- and, with `always_switch`, one
 * mc_lattice_switch attempt (mc_moves.F90:1536-1594; `npt` selects its ensemble branch).  Each walker
 * starts from the weight table given to mw_sweep_configure; the host synchronises the tables across
 * walkers and GPUs (comms_allreduce_eta/hist/uhist semantics) through get/set_tables. */
int mw_sweep_options(int record, int samplerun, int always_switch, int npt,
                     double av_binwidth, double wl_factor, double log_unbiased_norm, double pressure);
/* Volume moves inside the sweep (mc_volume, mc_moves.F90:1216-1534; ref_ljr not carried): per move the stream's
 * eighth number chooses a translation (< transP, mc_moves.F90:157-166,226-235) or a volume move: one symmetric
 * hmatrix element of both lattices changes by at most dv_max, every position is rescaled through fractional
 * coordinates, image vectors are rebuilt and the full-box energies recomputed with the existing lists, all on the
 * device; pressure and ensemble come from mw_sweep_options.  Default: translations only.  After sweeps with volume
 * moves call mw_sweep_sync_cells before rebuilding lists: it reads the cells back (h_out, count x 9, may be NULL)
 * and refreshes the host-side image vectors / neighbour-grid descriptors exactly as mw_set_cell does. */
/* leshift (userparams.f90:41; main.f90:146-150,173): the reference enthalpies of the two lattices enter the order
 * parameter (mc_moves.F90:1371,1526,1584) and the lattice-switch acceptance (:1567,1572).  (0, 0) = off. */
int mw_sweep_leshift(double ref_enthalpy_1, double ref_enthalpy_2);
/* The reference's -DMINU build as a run option (mc_moves.F90:1119-1140,1168-1170,1385-1401,1426-1429; off in every
 * shipped example): an accepted translation or volume move also takes the walker to the lattice with the lower
 * enthalpy (E + PV, minus ref_enthalpy under leshift), the acceptance carrying the switch's terms.  Two lattices only. */
int mw_sweep_minu(int on);
/* wl_swetnam (mc_moves.F90:1636-1653): after every recorded move the walker's increment becomes
 * min(orig_wl_factor, wl_alpha nbins log(rms deviation of its histogram from flat)); mu_min / mu_max of the whole grid. */
int mw_sweep_swetnam(int on, double wl_alpha, double orig_wl_factor, double mu_min, double mu_max);
/* parallel_strategy = 'dd' (mc_moves.F90:181-210,243-248,659-709): every walker samples its own window of the overlap
 * parameter (mw_sweep_windows; NULL arrays: back to the window of mw_sweep_configure for everybody).  Until cycle
 * eq_mc_cycles a walker outside its window carries no weight and no lattice switch is attempted; a walker still outside
 * at eq_mc_cycles raises the flag mw_sweep_check_flags reports.  Cycle numbers follow from move0 (nwater moves per cycle). */
int mw_sweep_dd(int on, int eq_mc_cycles);
int mw_sweep_windows(int first_walker, int count, const int *start_bin, const int *end_bin,
                     const double *mu_lo, const double *mu_hi);
/* Per-walker Wang-Landau increment and Swetnam visit total (every rank of the reference has its own; mw_sweep_options
 * sets one increment for all walkers unless 'dd' or wl_swetnam is on).  NULL pointers are skipped. */
int mw_sweep_set_factors(int first_walker, int count, const double *wl_factor, const double *sumhist);
/* Per-walker step sizes (bohr): every rank of the reference tunes its own mc_max_trans / mc_dv_max during equilibration
 * (eq_adjust_mc, mc_monitor_stats, mc_moves.F90:1724-1732).  NULL arrays: everybody back on the values of
 * mw_sweep_configure / mw_sweep_moves.  mw_sweep_get_counters: accepted translations, attempted and accepted volume
 * moves per walker since mw_sweep_configure (the host differences them per report interval). */
int mw_sweep_steps(int first_walker, int count, const double *max_trans_bohr, const double *dv_max_bohr);
int mw_sweep_get_counters(int first_walker, int count, long long *accepted, long long *vol_attempted, long long *vol_accepted);
int mw_sweep_get_factors(int first_walker, int count, double *wl_factor, double *sumhist, int *in_window);
int mw_sweep_moves(double transP, double dv_max_bohr);
int mw_sweep_get_volume_moves(int walker, long long *attempted, long long *accepted);
/* Nonzero (with the walker named in mw_last_error) if a volume move of any walker of the range needed more image
 * vectors than the table holds: that move was rejected and undone, the walker's state is consistent, but its chain
 * no longer follows mc_volume.  mw_sweep_moves reserves one extra shell of images, so this means a collapsing cell. */
int mw_sweep_check_flags(int first_walker, int count);
int mw_sweep_sync_cells(int first_ils, int count, double *h_out);
int mw_sweep_get_tables(int walker, double *weight, double *histogram, double *unbiased_hist);
int mw_sweep_set_tables(int walker, const double *weight, const double *histogram, const double *unbiased_hist);
/* The same for `count` consecutive walkers in one transfer: arrays of count x nbins doubles. */
int mw_sweep_get_tables_range(int first_walker, int count, double *weight, double *histogram, double *unbiased_hist);
int mw_sweep_set_tables_range(int first_walker, int count, const double *weight, const double *histogram,
                              const double *unbiased_hist);
int mw_sweep_get_switches(int walker, long long *switches);
/* mc_update_wl_bins subtracts the window minimum from a walker's weights after every update
 * (mc_moves.F90:1682-1685).  shifts[w] = the sum of those minima since the last call with reset != 0:
 * weight + shift is the walker's table without that gauge change, which is what a many-walker farm has to
 * sum (the reference's delta scheme, comms_mpi.f90:243-277, sums the gauge change of every rank as well). */
int mw_sweep_get_shifts_range(int first_walker, int count, double *shifts, int reset);
/* The sums of the exchange step for a farm, taken on the device (the tables never leave it): per table and bin,
 * sum_x[b] = sum over the walkers of the range of (table[w][b] (+ shift[w] for the weights when use_shifts) - last_x[b]),
 * in a fixed order.  A NULL last_x / sum_x skips that table.  reset_shifts zeroes the accumulated shifts afterwards.
 * mw_sweep_broadcast_tables writes one row (nbins doubles; NULL skips) into every walker of the range.
 * (comms_allreduce_eta/hist/uhist, comms_mpi.f90:243-277,461-530, with every walker a rank.) */
int mw_sweep_reduce_tables(int first_walker, int count, const double *last_w, const double *last_h, const double *last_u,
                           double *sum_w, double *sum_h, double *sum_u, int use_shifts, int reset_shifts);
int mw_sweep_broadcast_tables(int first_walker, int count, const double *weight, const double *histogram,
                              const double *unbiased_hist);
int mw_set_model_energy(int ils, double e);
int mw_sweep_set_state(int walker, int ls, double ls_mu);
/* The same for `count` consecutive walkers in two transfers (a farm of thousands sets up in a handful of calls). */
int mw_sweep_set_states_range(int first_walker, int count, const int *ls, const double *ls_mu);
int mw_sweep_get_state(int walker, int *ls, double *ls_mu, double *model_energy, long long *accepted);
/* nmoves moves for walkers first_walker .. first_walker+count-1.  log (may be NULL) receives
 * 8 doubles per walker and move: imol, accepted, old1, new1, old2, new2, ls_mu after, diffkT. */
int mw_sweep_translation(int first_walker, int count, int nmoves, unsigned long long seed,
                         unsigned long long move0, double *log);
int mw_sweep_translation_launch(int first_walker, int count, int nmoves, unsigned long long seed,
                                unsigned long long move0, int want_log);
/* What the last launch of the driver looked like (any pointer may be NULL): lattices per walker, moves in flight per walker
 * (look-ahead: 1, 2 or 4), where a walker's data lived (0: global memory, 1: positions in LDS, 2: positions and list rows in
 * LDS), whether the build carried volume moves, dynamic LDS per workgroup, and the LDS row stride (0 when rows stay global). */
int mw_sweep_last_launch(int *nlat, int *ahead, int *residency, int *volume_moves, int *lds_bytes, int *row_stride);
/* Dynamic LDS (bytes) of one walker's workgroup when its positions and list rows live in LDS (nwater <= 64), for list
 * rows of `row_stride` entries -- host arithmetic only, no device needed.  Eight walkers share a compute unit while this
 * plus the kernel's static LDS stays within 20480 bytes; the launch picks row_stride = the longest row of any box,
 * rounded up to even.  image_capacity: image vectors per box the engine has room for (0: 32 as after mw_init, or 48 with
 * volume moves -- mw_sweep_moves keeps room for one more shell of images than a 27-image cell has).  Returns -1 for
 * arguments no launch would use. */
int mw_sweep_lds_bytes(int nlat, int nwater, int nbins, int row_stride, int volume_moves, int samplerun, int image_capacity);

/* HIP-event timers on the engine's stream: slot in 0..4095. */
int mw_timer_start(int slot);
int mw_timer_stop(int slot);
int mw_timer_elapsed_ms(int slot, float *ms);    /* synchronises on the stop event */

/* Device facts for bench.py: name, CU count, total global memory bytes. */
int mw_device_info(char *name, int name_len, int *compute_units, long long *global_mem);

#ifdef __cplusplus
}
#endif
#endif /* MW_ENERGY_H */
