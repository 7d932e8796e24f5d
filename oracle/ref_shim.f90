! ORACLE / TEST INFRASTRUCTURE -- not product code.
!
! Thin bind(C) shim (our code) around the *unmodified* reference modules
! `constants`, `userparams`, `util`, `model` and `energy`
! (/root/reference/{constants,userparams,util,data_structures}.f90, molint.F90),
! which oracle/Makefile compiles from where they lie into oracle/_ref/.
! Nothing from the reference is copied here; this file only calls its public
! interface (molint.F90:22-37) so that Python (ctypes) can drive it to
!   * generate the golden vectors under tests/golden/ (tests/golden/make_golden.py)
!   * pin oracle/mw_oracle.c (the C restatement)
!   * serve as bench.py's cpu_baseline (kind "reference").
!
! G2 hazard (SURVEY.md section 0): compute_local_real_energy reads a stack
! array (vexplist, molint.F90:239) that it never writes for out-of-range
! slots (molint.F90:373-377, 385).  ref_local_energy therefore zeroes the
! stack region below its own frame (mw_scrub_stack, oracle/ref_scrub.c)
! immediately before every call, so 0*exp(garbage) is always exactly 0.
module mw_ref_shim
  use iso_c_binding
  implicit none
  interface
     subroutine mw_scrub_stack() bind(C, name="mw_scrub_stack")
     end subroutine mw_scrub_stack
  end interface
contains

  ! Create the model containers and run the reference's energy_init
  ! (molint.F90:91-153): ivects, neighbour lists and first energies.
  ! h_bohr is hmatrix(:,:,ils) column-major, xyz_bohr is ljr(:,1,:,ils).
  integer(c_int) function ref_init(nw, nlat, h_bohr, xyz_bohr) bind(C, name="ref_init")
    use userparams, only : nwater, num_lattices, model_type
    use model,      only : create_model, hmatrix, ljr
    use energy,     only : energy_init
    integer(c_int), value :: nw, nlat
    real(c_double), intent(in) :: h_bohr(3,3,nlat)
    real(c_double), intent(in) :: xyz_bohr(3,nw,nlat)
    integer :: ils, i
    nwater = nw
    num_lattices = nlat
    model_type = "mW"
    call create_model()
    do ils = 1, nlat
       hmatrix(:,:,ils) = h_bohr(:,:,ils)
       do i = 1, nw
          ljr(:,1,i,ils) = xyz_bohr(:,i,ils)
       end do
    end do
    call energy_init()
    ref_init = 0
  end function ref_init

  ! Release everything so that ref_init may be called again with another size.
  subroutine ref_finalize() bind(C, name="ref_finalize")
    use model,  only : destroy_model, ljr, ref_ljr
    use energy, only : energy_deinit, model_energy, nivect, nn, jn, vn
    call energy_deinit()
    if (allocated(model_energy)) deallocate(model_energy)
    if (allocated(nivect)) deallocate(nivect)
    if (allocated(nn)) deallocate(nn)
    if (allocated(jn)) deallocate(jn)
    if (allocated(vn)) deallocate(vn)
    call destroy_model()
    if (allocated(ljr)) deallocate(ljr)
    if (allocated(ref_ljr)) deallocate(ref_ljr)
  end subroutine ref_finalize

  subroutine ref_set_positions(ils, xyz_bohr) bind(C, name="ref_set_positions")
    use userparams, only : nwater
    use model,      only : ljr
    integer(c_int), value :: ils
    real(c_double), intent(in) :: xyz_bohr(3,nwater)
    integer :: i
    do i = 1, nwater
       ljr(:,1,i,ils) = xyz_bohr(:,i)
    end do
  end subroutine ref_set_positions

  subroutine ref_set_position(ils, imol, r) bind(C, name="ref_set_position")
    use model, only : ljr
    integer(c_int), value :: ils, imol
    real(c_double), intent(in) :: r(3)
    ljr(:,1,imol,ils) = r(:)
  end subroutine ref_set_position

  subroutine ref_get_positions(ils, xyz_bohr) bind(C, name="ref_get_positions")
    use userparams, only : nwater
    use model,      only : ljr
    integer(c_int), value :: ils
    real(c_double), intent(out) :: xyz_bohr(3,nwater)
    integer :: i
    do i = 1, nwater
       xyz_bohr(:,i) = ljr(:,1,i,ils)
    end do
  end subroutine ref_get_positions

  subroutine ref_set_cell(ils, h_bohr) bind(C, name="ref_set_cell")
    use model, only : hmatrix
    integer(c_int), value :: ils
    real(c_double), intent(in) :: h_bohr(3,3)
    hmatrix(:,:,ils) = h_bohr(:,:)
  end subroutine ref_set_cell

  ! compute_ivects (molint.F90:174-217); returns nivect(ils) and copies the
  ! vectors into out(3,maxout).
  integer(c_int) function ref_compute_ivects(ils, out, maxout) bind(C, name="ref_compute_ivects")
    use energy, only : compute_ivects, nivect, ivect
    integer(c_int), value :: ils, maxout
    real(c_double), intent(out) :: out(3,maxout)
    integer :: k
    call compute_ivects(ils)
    do k = 1, min(nivect(ils), maxout)
       out(:,k) = ivect(:,k,ils)
    end do
    ref_compute_ivects = nivect(ils)
  end function ref_compute_ivects

  ! compute_neighbours (molint.F90:501-559).
  subroutine ref_compute_neighbours(ils) bind(C, name="ref_compute_neighbours")
    use energy, only : compute_neighbours
    integer(c_int), value :: ils
    call compute_neighbours(ils)
  end subroutine ref_compute_neighbours

  ! Copy out nn(:,ils), jn(:,:,ils), vn(:,:,ils); jn_out/vn_out are (maxneigh, nwater).
  integer(c_int) function ref_get_neighbours(ils, nn_out, jn_out, vn_out) bind(C, name="ref_get_neighbours")
    use userparams, only : nwater
    use energy,     only : nn, jn, vn, maxneigh
    integer(c_int), value :: ils
    integer(c_int), intent(out) :: nn_out(nwater)
    integer(c_int), intent(out) :: jn_out(maxneigh,nwater), vn_out(maxneigh,nwater)
    integer :: i, k
    do i = 1, nwater
       nn_out(i) = nn(i,ils)
       do k = 1, maxneigh
          if (k <= nn(i,ils)) then
             jn_out(k,i) = jn(k,i,ils)
             vn_out(k,i) = vn(k,i,ils)
          else
             jn_out(k,i) = 0
             vn_out(k,i) = 0
          end if
       end do
    end do
    ref_get_neighbours = maxneigh
  end function ref_get_neighbours

  ! compute_model_energy (molint.F90:407-499) -> model_energy(ils).
  real(c_double) function ref_model_energy(ils) bind(C, name="ref_model_energy")
    use energy, only : compute_model_energy, model_energy
    integer(c_int), value :: ils
    call compute_model_energy(ils)
    ref_model_energy = model_energy(ils)
  end function ref_model_energy

  ! compute_local_real_energy (molint.F90:220-404), with the G2 stack scrub.
  real(c_double) function ref_local_energy(imol, ils) bind(C, name="ref_local_energy")
    use energy, only : compute_local_real_energy
    integer(c_int), value :: imol, ils
    call mw_scrub_stack()
    ref_local_energy = compute_local_real_energy(imol, ils)
  end function ref_local_energy

  ! All local energies of lattice ils (scrubbed before each call).
  subroutine ref_local_energy_all(ils, e_out) bind(C, name="ref_local_energy_all")
    use userparams, only : nwater
    use energy,     only : compute_local_real_energy
    integer(c_int), value :: ils
    real(c_double), intent(out) :: e_out(nwater)
    integer :: i
    do i = 1, nwater
       call mw_scrub_stack()
       e_out(i) = compute_local_real_energy(i, ils)
    end do
  end subroutine ref_local_energy_all

  ! Trial-move records: for each move m, atom imol(m) of lattice ils is set to
  ! trial(:,m), the local energy evaluated, and the atom put back -- the
  ! old/new pattern of mc_water_translation (mc_moves.F90:1010,1079-1083,1186).
  subroutine ref_trial_moves(ils, nmoves, imol, trial, e_old, e_new) bind(C, name="ref_trial_moves")
    use model,  only : ljr
    use energy, only : compute_local_real_energy
    integer(c_int), value :: ils, nmoves
    integer(c_int), intent(in) :: imol(nmoves)
    real(c_double), intent(in) :: trial(3,nmoves)
    real(c_double), intent(out) :: e_old(nmoves), e_new(nmoves)
    real(c_double) :: keep(3)
    integer :: m
    do m = 1, nmoves
       call mw_scrub_stack()
       e_old(m) = compute_local_real_energy(imol(m), ils)
       keep(:) = ljr(:,1,imol(m),ils)
       ljr(:,1,imol(m),ils) = trial(:,m)
       call mw_scrub_stack()
       e_new(m) = compute_local_real_energy(imol(m), ils)
       ljr(:,1,imol(m),ils) = keep(:)
    end do
  end subroutine ref_trial_moves

  ! The model constants as the compiled reference holds them (molint.F90:63-74):
  ! out = sigma, epsilon, lambda, A, B, gamma, a, cos0.
  subroutine ref_constants(out) bind(C, name="ref_constants")
    use energy, only : mw_sigma, mw_epsilon, mw_lambda, sw_bigA, sw_B, sw_gamma, sw_a, cos0
    real(c_double), intent(out) :: out(8)
    out = (/ mw_sigma, mw_epsilon, mw_lambda, sw_bigA, sw_B, sw_gamma, sw_a, cos0 /)
  end subroutine ref_constants

  ! Timing legs for bench.py's cpu_baseline: nrep back-to-back evaluations.
  real(c_double) function ref_time_model_energy(ils, nrep) bind(C, name="ref_time_model_energy")
    use energy, only : compute_model_energy, model_energy
    integer(c_int), value :: ils, nrep
    integer :: r
    real(c_double) :: acc
    acc = 0.0_c_double
    do r = 1, nrep
       call compute_model_energy(ils)
       acc = acc + model_energy(ils)
    end do
    ref_time_model_energy = acc
  end function ref_time_model_energy

  real(c_double) function ref_time_local_energy(ils, nrep, ncalls, imol) bind(C, name="ref_time_local_energy")
    use energy, only : compute_local_real_energy
    integer(c_int), value :: ils, nrep, ncalls
    integer(c_int), intent(in) :: imol(ncalls)
    integer :: r, m
    real(c_double) :: acc
    acc = 0.0_c_double
    call mw_scrub_stack()
    do r = 1, nrep
       do m = 1, ncalls
          acc = acc + compute_local_real_energy(imol(m), ils)
       end do
    end do
    ref_time_local_energy = acc
  end function ref_time_local_energy

end module mw_ref_shim
