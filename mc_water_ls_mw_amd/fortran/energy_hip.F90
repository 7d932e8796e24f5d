!=============================================================================!
!                     E N E R G Y   (MI355X / HIP engine)                     !
!=============================================================================!
! Drop-in replacement for the reference's `module energy` (molint.F90): same   !
! module name, same public routines, argument lists and public variables       !
! (molint.F90:22-37,47-48), so main.f90 / mc_moves.F90 / io.f90 compile and    !
! link against it unchanged.  Every routine forwards through ISO_C_BINDING to   !
! libmw_hip.so (include/mw_energy.h); nothing is computed on the CPU here.      !
!                                                                               !
! Implicit inputs, exactly as in the reference: model::ljr, model::hmatrix,     !
! model::volume (written by energy_init), userparams::nwater/num_lattices.      !
! model_energy stays an ordinary host array because mc_moves writes it          !
! directly (mc_moves.F90:1013-1016,1087,1190; SURVEY.md G10).                   !
!                                                                               !
! Host <-> device coherence (the callers mutate ljr without telling us):        !
!   compute_neighbours / compute_model_energy : re-mirror all positions of      !
!       the lattice (these follow every bulk change: volume move                !
!       mc_moves.F90:1314-1357, chain sync :2331-2396, restart :842-852)        !
!   compute_local_real_energy(imol,ils) : sends the host position of imol AND   !
!       of the molecule queried just before it in that lattice (whose trial     !
!       move may have been silently reverted, mc_moves.F90:1186)                !
!   compute_ivects : re-mirrors the cell AND marks the lattice's positions stale: !
!       a rejected volume move rescales every position back on the host and     !
!       then calls nothing but compute_ivects (mc_moves.F90:1410-1514), so the   !
!       next local-energy call re-mirrors all positions first.                   !
!=============================================================================!
module energy

  use iso_c_binding
  use constants, only : dp,int32,ang_to_bohr

  implicit none
  private

  public :: energy_init
  public :: energy_deinit
  public :: compute_ivects
  public :: compute_model_energy
  public :: compute_local_real_energy
  public :: compute_neighbours

  public :: model_energy
  public :: nivect,ivect
  public :: maxneigh
  public :: nn,jn,vn
  public :: energy_fetch_neighbours   ! extension: fill nn/jn/vn from the device list

  ! current energy due to the model Hamiltonian (host array, written by callers too)
  real(kind=dp),allocatable,dimension(:),save :: model_energy

  ! lattice translation vectors (host copies of what the device holds)
  integer,allocatable,dimension(:) :: nivect
  real(kind=dp),allocatable,dimension(:,:,:) :: ivect

  public :: mw_sigma,mw_epsilon,mw_lambda
  public :: sw_bigA,sw_B,sw_gamma,sw_a,sw_p,sw_q,cos0

  ! Model constants, kept as parameters for source compatibility (molint.F90:64-74).
  ! The engine holds its own copies; energy_init checks that the two agree bit for bit.
  real(kind=dp),parameter :: mw_sigma   = 2.3925_dp*ang_to_bohr
  real(kind=dp),parameter :: mw_epsilon = 6.189_dp/627.509469_dp
  real(kind=dp),parameter :: mw_lambda  = 23.15_dp
  real(kind=dp),parameter :: sw_bigA = 7.049556277_dp
  real(kind=dp),parameter :: sw_B = 0.6022245584_dp
  real(kind=dp),parameter :: sw_gamma = 1.2_dp
  real(kind=dp),parameter :: sw_a = 1.8_dp
  integer,parameter :: sw_p=4,sw_q=0
  ! single-precision literal widened to double, as in the reference (SURVEY.md G1)
  real(kind=dp),parameter :: cos0 = real(-0.33331324756,kind=dp)

  ! Neighbour list.  The device owns the list; nn/jn/vn keep their names and shapes
  ! and are refreshed by energy_fetch_neighbours (nothing outside the module reads them).
  integer,parameter :: maxneigh = 50
  integer,allocatable,dimension(:,:),save :: nn
  integer,allocatable,dimension(:,:,:),save :: jn,vn

  ! molecule queried last in each lattice (0 = none since the last full mirror)
  integer,allocatable,dimension(:),save :: last_imol
  ! the request posted ahead for lattice 2 (see compute_local_real_energy): what it carried
  logical,save :: pend = .false.,overlap_calls = .true.
  integer,save :: pend_imol = 0
  integer(c_int),save :: pend_prev = 0
  real(c_double),save :: pend_r1(3) = 0.0_dp,pend_r2(3) = 0.0_dp
  ! .true. after compute_ivects until the lattice's positions have been mirrored again
  logical,allocatable,dimension(:),save :: stale

  interface
     integer(c_int) function mw_init(device,nwater,nboxes,maxn) bind(C,name="mw_init")
       import :: c_int
       integer(c_int),value :: device,nwater,nboxes,maxn
     end function mw_init
     integer(c_int) function mw_finalize() bind(C,name="mw_finalize")
       import :: c_int
     end function mw_finalize
     type(c_ptr) function mw_last_error() bind(C,name="mw_last_error")
       import :: c_ptr
     end function mw_last_error
     integer(c_int) function mw_constants(out) bind(C,name="mw_constants")
       import :: c_int,c_double
       real(c_double),intent(out) :: out(8)
     end function mw_constants
     integer(c_int) function mw_set_cell(ils,h,nivect_out) bind(C,name="mw_set_cell")
       import :: c_int,c_double
       integer(c_int),value :: ils
       real(c_double),intent(in) :: h(9)
       integer(c_int),intent(out) :: nivect_out
     end function mw_set_cell
     integer(c_int) function mw_get_ivects(ils,out,max_vectors,nivect_out) bind(C,name="mw_get_ivects")
       import :: c_int,c_double
       integer(c_int),value :: ils,max_vectors
       real(c_double),intent(out) :: out(*)
       integer(c_int),intent(out) :: nivect_out
     end function mw_get_ivects
     integer(c_int) function mw_upload_positions(ils,xyz) bind(C,name="mw_upload_positions")
       import :: c_int,c_double
       integer(c_int),value :: ils
       real(c_double),intent(in) :: xyz(*)
     end function mw_upload_positions
     integer(c_int) function mw_build_neighbours(ils,min_nn,max_nn) bind(C,name="mw_build_neighbours")
       import :: c_int
       integer(c_int),value :: ils
       integer(c_int),intent(out) :: min_nn,max_nn
     end function mw_build_neighbours
     integer(c_int) function mw_get_neighbours(ils,nn_out,jn_out,vn_out) bind(C,name="mw_get_neighbours")
       import :: c_int
       integer(c_int),value :: ils
       integer(c_int),intent(out) :: nn_out(*),jn_out(*),vn_out(*)
     end function mw_get_neighbours
     integer(c_int) function mw_model_energy(ils,e) bind(C,name="mw_model_energy")
       import :: c_int,c_double
       integer(c_int),value :: ils
       real(c_double),intent(out) :: e
     end function mw_model_energy
     integer(c_int) function mw_model_energy_of(ils,xyz,e) bind(C,name="mw_model_energy_of")
       import :: c_int,c_double
       integer(c_int),value :: ils
       real(c_double),intent(in) :: xyz(3,*)
       real(c_double),intent(out) :: e
     end function mw_model_energy_of
     integer(c_int) function mw_local_energy_patched(ils,imol,r_imol,imol_prev,r_prev,e) &
          bind(C,name="mw_local_energy_patched")
       import :: c_int,c_double
       integer(c_int),value :: ils,imol,imol_prev
       real(c_double),intent(in) :: r_imol(3),r_prev(3)
       real(c_double),intent(out) :: e
     end function mw_local_energy_patched
     integer(c_int) function mw_local_energy_post(ils,imol,r_imol,imol_prev,r_prev) bind(C,name="mw_local_energy_post")
       import :: c_int,c_double
       integer(c_int),value :: ils,imol,imol_prev
       real(c_double),intent(in) :: r_imol(3),r_prev(3)
     end function mw_local_energy_post
     integer(c_int) function mw_local_energy_collect(ils,e) bind(C,name="mw_local_energy_collect")
       import :: c_int,c_double
       integer(c_int),value :: ils
       real(c_double),intent(out) :: e
     end function mw_local_energy_collect
     integer(c_size_t) function c_strlen(s) bind(C,name="strlen")
       import :: c_size_t,c_ptr
       type(c_ptr),value :: s
     end function c_strlen
  end interface

contains

  subroutine mw_check(rc,where)
    !------------------------------------------------------------------------------!
    ! The reference has no status codes: failures are fatal `stop`s                !
    ! (molint.F90:109-143,167).  A nonzero return from the engine is fatal too.    !
    !------------------------------------------------------------------------------!
    integer(c_int),intent(in) :: rc
    character(len=*),intent(in) :: where
    type(c_ptr) :: p
    character(kind=c_char),pointer :: msg(:)
    integer :: n,k
    if (rc==0) return
    p = mw_last_error()
    n = int(c_strlen(p))
    call c_f_pointer(p,msg,(/n/))
    write(0,'("Error in ",A," : ")',advance='no')where
    do k = 1,n
       write(0,'(A1)',advance='no')msg(k)
    end do
    write(0,*)
    stop 'Error in mW HIP energy engine'
  end subroutine mw_check

  subroutine energy_init
    !------------------------------------------------------------------------------!
    ! As molint.F90:91-153: allocate, set volume(ils), build image vectors, lists  !
    ! and first energies for every lattice.  The device is chosen from the local   !
    ! rank (device = -1: MW_DEVICE / LOCAL_RANK / OMPI_COMM_WORLD_LOCAL_RANK /     !
    ! SLURM_LOCALID, modulo the device count).                                     !
    !------------------------------------------------------------------------------!
    use util,       only : util_determinant
    use userparams, only : num_lattices,nwater
    use model,      only : hmatrix,volume
    implicit none
    integer :: ils,im,jm,km,ierr
    real(c_double) :: cdev(8)
    character(len=8) :: envval

    allocate(model_energy(1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating model and recip energy arrays'
    allocate(nivect(1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating nivect'
    allocate(last_imol(1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating last_imol'
    last_imol = 0
    pend = .false.
    call get_environment_variable('MW_LOCAL_OVERLAP',envval,status=ierr)
    overlap_calls = .not.(ierr==0 .and. envval(1:1)=='0')
    allocate(stale(1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating stale flags'
    stale = .true.

    call mw_check(mw_init(-1_c_int,int(nwater,c_int),int(num_lattices,c_int),int(maxneigh,c_int)),'energy_init')

    ! the constants compiled into the engine must be the reference's, bit for bit
    call mw_check(mw_constants(cdev),'energy_init')
    if ( cdev(1)/=mw_sigma .or. cdev(2)/=mw_epsilon .or. cdev(3)/=mw_lambda .or. cdev(4)/=sw_bigA .or. &
         cdev(5)/=sw_B .or. cdev(6)/=sw_gamma .or. cdev(7)/=sw_a .or. cdev(8)/=cos0 ) then
       stop 'Error in energy_init : engine constants differ from module parameters'
    end if

    do ils = 1,num_lattices
       ! same padded estimate as molint.F90:117-121 for the size of the public ivect array
       im = floor((sw_a*mw_sigma+1.0_dp)/sqrt(dot_product(hmatrix(:,1,ils),hmatrix(:,1,ils))))+1
       jm = floor((sw_a*mw_sigma+1.0_dp)/sqrt(dot_product(hmatrix(:,2,ils),hmatrix(:,2,ils))))+1
       km = floor((sw_a*mw_sigma+1.0_dp)/sqrt(dot_product(hmatrix(:,3,ils),hmatrix(:,3,ils))))+1
       nivect(ils) = (2*im+1)*(2*jm+1)*(2*km+1)
       volume(ils) = abs(util_determinant(hmatrix(:,:,ils)))          ! molint.F90:125
    end do

    allocate(ivect(1:3,1:maxval(nivect,1),1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating ivect'

    do ils = 1,num_lattices
       call compute_ivects(ils)
    end do

    allocate(nn(1:nwater,1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating nn array in molint.F90'
    allocate(vn(1:maxneigh,1:nwater,1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating vn array in molint.F90'
    allocate(jn(1:maxneigh,1:nwater,1:num_lattices),stat=ierr)
    if (ierr/=0) stop 'Error allocating jn array in molint.F90'
    nn = 0 ; jn = 0 ; vn = 0

    do ils = 1,num_lattices
       call compute_neighbours(ils)
       call compute_model_energy(ils)
    end do

    return

  end subroutine energy_init

  subroutine energy_deinit()
    !------------------------------------------------------------------------------!
    ! As molint.F90:155-171 (which frees only ivect), plus the device state.       !
    !------------------------------------------------------------------------------!
    implicit none
    integer :: ierr
    deallocate(ivect,stat=ierr)
    if (ierr/=0) stop 'Error deallocating ivect'
    call mw_check(mw_finalize(),'energy_deinit')
    return
  end subroutine energy_deinit

  subroutine compute_ivects(ils)
    !------------------------------------------------------------------------------!
    ! As molint.F90:174-217: image translation vectors of lattice ils from         !
    ! model::hmatrix, mirrored on the device; nivect/ivect refreshed on the host.  !
    !------------------------------------------------------------------------------!
    use model, only : hmatrix
    implicit none
    integer,intent(in) :: ils
    integer(c_int) :: n
    real(kind=dp),allocatable,dimension(:,:,:) :: tmp
    real(c_double) :: h9(9)
    integer :: ierr

    h9 = reshape(hmatrix(:,:,ils),(/9/))
    call mw_check(mw_set_cell(int(ils,c_int),h9,n),'compute_ivects')
    nivect(ils) = n
    stale(ils) = .true.        ! cell changes come with bulk position changes (volume move and its rejection)
    if (allocated(ivect)) then
       if (n>size(ivect,2)) then
          ! the reference would overrun here when the cell shrinks (SURVEY.md G9); grow instead
          allocate(tmp(1:3,1:n,1:size(ivect,3)),stat=ierr)
          if (ierr/=0) stop 'Error allocating ivect'
          tmp = 0.0_dp
          tmp(:,1:size(ivect,2),:) = ivect
          call move_alloc(tmp,ivect)
       end if
       call mw_check(mw_get_ivects(int(ils,c_int),ivect(:,:,ils),int(size(ivect,2),c_int),n),'compute_ivects')
    end if
    return
  end subroutine compute_ivects

  real(kind=dp) function compute_local_real_energy(imol,ils)
    !------------------------------------------------------------------------------!
    ! As molint.F90:220-404: energy of every pair and triplet involving imol,      !
    ! from the host's CURRENT ljr (see the coherence note at the top).             !
    !------------------------------------------------------------------------------!
    use model,      only : ljr
    use userparams, only : num_lattices
!$  use omp_lib,    only : omp_in_parallel
    implicit none
    integer,intent(in) :: imol,ils
    real(c_double) :: e,r1(3),r2(3)
    integer(c_int) :: prev,rc
    logical :: hit,par

    ! Inside an OpenMP parallel region (the reference's dormant `!$omp parallel do` over ils, mc_moves.F90:1006-1018) the
    ! two lattices are asked about by two threads at once: nothing is posted ahead there -- the threads ARE the overlap --
    ! and the bookkeeping of a request posted earlier by serial code is settled by one thread at a time.
    par = .false.
!$  par = omp_in_parallel()
!$omp critical (mw_energy_pending)
    ! The answer may be waiting already: the host asks the same question of lattice 2 right after lattice 1
    ! (mc_moves.F90:1006-1018, 1076-1092), so the call for lattice 1 posted lattice 2's request before it waited for
    ! its own reply (below).  The reply is this call's if nothing the request carried has changed since: the
    ! molecule, its position, the previously queried molecule's position, and no bulk change of the lattice.
    if (pend) then
       pend = .false.
       hit = ils==2 .and. imol==pend_imol .and. .not.stale(2)
       if (hit) hit = all(ljr(:,1,imol,2)==pend_r1)
       if (hit .and. pend_prev>=1) hit = all(ljr(:,1,pend_prev,2)==pend_r2)
       rc = mw_local_energy_collect(2_c_int,e)            ! (a reply nobody wants is still waited for: the slot holds one request)
       if (rc/=0 .and. rc/=2) call mw_check(rc,'compute_local_real_energy')
       if (.not.(hit .and. rc==0)) hit = .false.
    else
       hit = .false.
    end if
!$omp end critical (mw_energy_pending)
    if (hit) then
       compute_local_real_energy = e                        ! (last_imol(2) = imol since the post)
       return
    end if

    if (stale(ils)) then
       call mw_check(mw_upload_positions(int(ils,c_int),ljr(:,1,:,ils)),'compute_local_real_energy')
       stale(ils) = .false.
       last_imol(ils) = 0
    end if
    ! Only the FIRST question about a molecule is worth posting ahead: the host asks for the old energies of both lattices
    ! back to back, but it moves the molecule lattice by lattice in between the two "new energy" calls (mc_moves.F90:1076-
    ! 1083), so lattice 2's trial position does not exist yet when lattice 1's is evaluated.
    if (ils==1 .and. num_lattices==2 .and. overlap_calls .and. .not.par .and. last_imol(1)/=imol) then
       if (.not.stale(2)) then
          r1 = ljr(:,1,imol,2)
          prev = last_imol(2)
          if (prev>=1 .and. prev/=imol) then
             r2 = ljr(:,1,prev,2)
          else
             prev = 0
             r2 = 0.0_dp
          end if
          rc = mw_local_energy_post(2_c_int,int(imol,c_int),r1,prev,r2)
          if (rc==0) then
             pend = .true. ; pend_imol = imol ; pend_prev = prev ; pend_r1 = r1 ; pend_r2 = r2
             last_imol(2) = imol
          else if (rc==2) then
             overlap_calls = .false.                        ! the resident server is switched off: nothing to overlap with
          else
             call mw_check(rc,'compute_local_real_energy')
          end if
       end if
    end if
    r1 = ljr(:,1,imol,ils)
    prev = last_imol(ils)
    if (prev>=1 .and. prev/=imol) then
       r2 = ljr(:,1,prev,ils)
    else
       prev = 0
       r2 = 0.0_dp
    end if
    call mw_check(mw_local_energy_patched(int(ils,c_int),int(imol,c_int),r1,prev,r2,e),'compute_local_real_energy')
    last_imol(ils) = imol
    compute_local_real_energy = e
    return
  end function compute_local_real_energy

  subroutine compute_model_energy(ils)
    !------------------------------------------------------------------------------!
    ! As molint.F90:407-499: full-box energy of lattice ils -> model_energy(ils).   !
    !------------------------------------------------------------------------------!
    use model, only : ljr
    implicit none
    integer,intent(in) :: ils
    real(c_double) :: e
    call mw_check(mw_model_energy_of(int(ils,c_int),ljr(:,1,:,ils),e),'compute_model_energy')   ! mirror + evaluate, one call
    last_imol(ils) = 0
    stale(ils) = .false.
    model_energy(ils) = e
    return
  end subroutine compute_model_energy

  subroutine compute_neighbours(ils)
    !------------------------------------------------------------------------------!
    ! As molint.F90:501-559: rebuild the Verlet list of lattice ils (same set of   !
    ! (jmol,image) entries in the same order), warn below 16 neighbours.           !
    !------------------------------------------------------------------------------!
    use userparams, only : nwater
    use model, only      : ljr
    implicit none
    integer,intent(in) :: ils
    integer(c_int) :: mn,mx
    integer :: imol

    call compute_ivects(ils)                                           ! molint.F90:518
    call mw_check(mw_upload_positions(int(ils,c_int),ljr(:,1,:,ils)),'compute_neighbours')
    last_imol(ils) = 0
    stale(ils) = .false.
    call mw_check(mw_build_neighbours(int(ils,c_int),mn,mx),'compute_neighbours')
    if (mn<16) then                                                    ! molint.F90:552-554
       call energy_fetch_neighbours(ils)
       do imol = 1,nwater
          if (nn(imol,ils) < 16 ) then
             write(0,'("WARNING: Molecule ",I5," has only ",I5," neighbours")')imol,nn(imol,ils)
          end if
       end do
    end if
    return
  end subroutine compute_neighbours

  subroutine energy_fetch_neighbours(ils)
    !------------------------------------------------------------------------------!
    ! Copy the device list of lattice ils into nn/jn/vn (reference layout).        !
    !------------------------------------------------------------------------------!
    implicit none
    integer,intent(in) :: ils
    call mw_check(mw_get_neighbours(int(ils,c_int),nn(:,ils),jn(:,:,ils),vn(:,:,ils)),'energy_fetch_neighbours')
    return
  end subroutine energy_fetch_neighbours

end module energy
