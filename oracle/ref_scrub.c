/* ORACLE / TEST INFRASTRUCTURE -- not product code.
 *
 * Stack scrub for the G2 hazard of the reference's compute_local_real_energy
 * (molint.F90:239 declares six 100-double stack arrays; vexplist is read at
 * molint.F90:385 for slots it never wrote).  Called by oracle/ref_shim.f90
 * right before every reference call, from the same stack depth, so the frame
 * the reference routine is about to occupy holds +0.0 everywhere.
 */
#include <stddef.h>

#define MW_SCRUB_DOUBLES 8192 /* 64 KiB >> the 4.8 KiB of scratch arrays */

__attribute__((noinline)) void mw_scrub_stack(void)
{
    volatile double pad[MW_SCRUB_DOUBLES];
    for (size_t i = 0; i < MW_SCRUB_DOUBLES; ++i) pad[i] = 0.0;
}
