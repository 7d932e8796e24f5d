/* ORACLE / TEST INFRASTRUCTURE -- not product code.
 *
 * Link-time interposition (ld --wrap) for the whole-program reference build
 * oracle/_ref/mc_water_ref: every call that the reference's mc_moves.o makes to
 * compute_local_real_energy (symbol _QMenergyPcompute_local_real_energy, flang ABI:
 * REAL(8) function, two INTEGER arguments by reference) first zeroes the stack below,
 * then runs the reference's own, unmodified routine.  Without it the reference reads
 * an uninitialised stack array (molint.F90:239,385): whenever the stale
 * bytes decode to a huge double, 0*exp(garbage) is NaN, the move's energy change is NaN,
 * the acceptance test compares false and the move is silently rejected -- the program
 * stays self-consistent but its trajectory depends on stack garbage.  With the scrub the
 * routine computes what its author intended (an out-of-range slot contributes 0).
 */
void mw_scrub_stack(void);
double __real__QMenergyPcompute_local_real_energy(int *imol, int *ils);

double __wrap__QMenergyPcompute_local_real_energy(int *imol, int *ils)
{
    mw_scrub_stack();
    return __real__QMenergyPcompute_local_real_energy(imol, ils);
}
