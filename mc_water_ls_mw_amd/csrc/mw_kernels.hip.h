// mw_kernels.hip.h -- gfx950 (MI355X, CDNA4) device code of the mW energy engine.
//
// What each kernel stands behind in the reference (keb721/mc_water_ls_mw):
//   k_build_neighbours  compute_neighbours       molint.F90:501-559
//   k_model_energy      compute_model_energy     molint.F90:407-499
//   k_move_energy / k_local_energy_single   compute_local_real_energy molint.F90:220-404
// None of it is a translation: the list is slot-major and packed for coalesced
// reads, positions of a whole box are staged in LDS, the three-body sum is
// evaluated from per-atom moments in O(neighbours), and the single-move path
// maps one request onto one 64-wide wavefront.  Double precision throughout;
// this is gather + transcendental work, so no MFMA.
//
// The device code lives in five headers, included here in dependency order:
//   mw_common.hip.h       constants, packed list entry, fp64 primitives, wave/DPP reductions
//   mw_neighbours.hip.h   neighbour-list builders
//   mw_full_energy.hip.h  full-box energy
//   mw_move_energy.hip.h  local energy / fused trial-move energy
//   mw_sweep.hip.h        device-resident Monte Carlo driver
#pragma once

#include "mw_common.hip.h"
#include "mw_neighbours.hip.h"
#include "mw_full_energy.hip.h"
#include "mw_move_energy.hip.h"
#include "mw_sweep.hip.h"
