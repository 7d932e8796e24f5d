"""MI355X-native mW (monatomic-water Stillinger-Weber) energy engine.

Drop-in for the hot path of keb721/mc_water_ls_mw's Fortran ``module energy``
(molint.F90): image vectors, Verlet neighbour list, full-box energy and
single-molecule local energy, as hand-written gfx950 HIP kernels behind the
C ABI declared in ``include/mw_energy.h``.
"""
__version__ = "0.1.0"
