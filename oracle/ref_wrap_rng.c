/* ORACLE / TEST INFRASTRUCTURE -- not product code.
 *
 * Link-time interposition (ld --wrap) of the reference's random_uniform_random
 * (symbol _QMrandomPrandom_uniform_random, called from mc_moves.o) for the build
 * oracle/_ref/mc_water_ref_rng: the reference's unmodified mc_cycle / mc_water_translation then
 * draw the SAME counter-based stream as oracle/mw_oracle.c's mwo_sweep_translation and the device
 * driver, so the oracle of the translation-move driver can be pinned against the reference program
 * itself (tests/test_sweep_pin.py).  mc_cycle draws one number to choose the move type
 * (mc_moves.F90:226) and mc_water_translation six (:1001,1021-1023,1035,1145): call c belongs to move
 * c/7; slot 0 returns 0 (always "translation" -- the test switches volume moves off),
 * slots 1..6 are the six numbers mwo_move_uniforms(seed, walker 0, move) defines.  With
 * mc_always_switch every move is followed by one mc_lattice_switch attempt that draws an eighth number
 * (mc_moves.F90:1576): run with MW_WRAP_CALLS_PER_MOVE=8, slot 7 is u[6] of mwo_move_uniforms8.
 * The reference's own start-up self-test of its generator calls the function inside random.o and is
 * not affected.
 */
#include "mw_oracle.h"
#include <stdlib.h>

#define MW_WRAP_SEED 424242ULL

/* State machine over the reference's call order inside one mc_cycle move (mc_moves.F90:224-250):
 *   1 draw  xi, the move type            -> u[7]; translation if xi < transP, else volume move
 *   translation: 6 draws (:1001,1021-1023,1035,1145) -> u[0..5]
 *   volume move: 4 draws (:1269,1271,1274,1378)      -> u[0..3]
 *   with mc_always_switch: 1 draw in mc_lattice_switch (:1576) -> u[6]
 * Environment: MW_WRAP_TRANSP = transP of the run (default 1: translations only),
 *              MW_WRAP_SWITCH = 1 if mc_always_switch (MW_WRAP_CALLS_PER_MOVE=8 is accepted as a synonym). */
static unsigned long long mw_move = 0;
static int mw_phase = -1;           /* -1: not configured; 0: expect xi; 1..: index into the move's draws */
static int mw_ndraws = 0, mw_switch = 0;
static unsigned long long mw_switch_from = 0;   /* MW_WRAP_SWITCH_FROM_MOVE: first move that is followed by a switch attempt
                                                   ('dd' runs attempt none while mc_cycle_num < eq_mc_cycles, mc_moves.F90:243-248) */
static double mw_transp = 1.0, mw_u[8];

double __wrap__QMrandomPrandom_uniform_random(void)
{
    if (mw_phase < 0) {
        const char *e = getenv("MW_WRAP_TRANSP");
        if (e) mw_transp = atof(e);
        e = getenv("MW_WRAP_SWITCH");
        if (e && atoi(e) == 1) mw_switch = 1;
        e = getenv("MW_WRAP_CALLS_PER_MOVE");
        if (e && atoi(e) == 8) mw_switch = 1;
        e = getenv("MW_WRAP_SWITCH_FROM_MOVE");
        if (e) mw_switch_from = strtoull(e, 0, 10);
        mw_phase = 0;
    }
    if (mw_phase == 0) {
        mwo_move_uniforms8(MW_WRAP_SEED, 0u, mw_move, mw_u);
        mw_ndraws = (mw_u[7] < mw_transp) ? 6 : 4;
        mw_phase = 1;
        return mw_u[7];
    }
    double v;
    if (mw_phase <= mw_ndraws) v = mw_u[mw_phase - 1];
    else v = mw_u[6];                                   /* the lattice-switch variate */
    if (mw_phase == mw_ndraws + ((mw_switch && mw_move >= mw_switch_from) ? 1 : 0)) { mw_phase = 0; ++mw_move; }
    else ++mw_phase;
    return v;
}

/* G11: the serial comms flavour never sets comms::size (comms_serial.f90:21; BSS, so 0 -- mc.log reports
 * "Number of MPI tasks : 0"), and mc_init's log_unbiased_norm multiplies by it (mc_moves.F90:781-782):
 * log(0) makes the unbiased histogram and delta G of a serial sample run NaN.  The MPI flavour has
 * size >= 1.  MW_WRAP_SIZE=n gives this build the value an n-rank MPI run would see (used with n = 1 by
 * tests/test_schedule_pin.py to pin log_unbiased_norm and mc_compute_deltaG_from_hist); unset, the
 * reference's own serial behaviour is left alone. */
extern int _QMcommsEsize;
extern int _QMcommsEmyrank;
__attribute__((constructor)) static void mw_wrap_comms_size(void)
{
    const char *e = getenv("MW_WRAP_SIZE");
    if (e && atoi(e) > 0) _QMcommsEsize = atoi(e);
}

/* MW_WRAP_RANK=r (with MW_WRAP_SIZE=R): from mc_init on the serial program behaves as rank r of an R-rank run wherever
 * it only consults comms::myrank / comms::size -- the 'dd' window assignment, the choice of the active lattice, the
 * equilibration guard and the per-window flatness check (mc_moves.F90:181-210,659-709,2002-2016).  The input files are
 * read by rank 0 only and "broadcast" (io.f90:106-145, init.f90:70-98), so the rank is poked AFTER that: energy_init
 * (main.f90:115) is the last call before mc_init, and it is interposed here just to carry the poke.  The collectives
 * stay the serial stubs, so nothing that needs another rank's data (the window joins) is pinned this way. */
void __real__QMenergyPenergy_init(void);
void __wrap__QMenergyPenergy_init(void)
{
    __real__QMenergyPenergy_init();
    const char *e = getenv("MW_WRAP_RANK");
    if (e && atoi(e) >= 0) _QMcommsEmyrank = atoi(e);
}
