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

static unsigned long long mw_wrap_calls = 0;

double __wrap__QMrandomPrandom_uniform_random(void)
{
    static unsigned long long per_move = 0;
    if (per_move == 0) {
        const char *e = getenv("MW_WRAP_CALLS_PER_MOVE");
        per_move = (e && atoi(e) == 8) ? 8ULL : 7ULL;
    }
    const unsigned long long c = mw_wrap_calls++;
    const unsigned long long move = c / per_move;
    const int slot = (int)(c % per_move);
    double u[8];
    if (slot == 0) return 0.0;
    mwo_move_uniforms8(MW_WRAP_SEED, 0u, move, u);
    return u[slot - 1];
}
