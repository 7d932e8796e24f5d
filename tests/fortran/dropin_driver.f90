! Test driver (our code) for the drop-in boundary: it `use`s module energy exactly
! the way the reference host does -- the call pattern of mc_water_translation
! (mc_moves.F90:1010-1190: old local energy, move ljr on the host, new local
! energy, accept or SILENTLY revert, model_energy edited by the caller), the list
! refresh of mc_cycle (mc_moves.F90:218-222), the monitor drift check
! (mc_moves.F90:1786-1792) and a volume move (mc_moves.F90:1285-1357: rescale
! hmatrix and all positions, compute_ivects, compute_model_energy, no list rebuild).
! It is compiled twice against the reference's own constants/userparams/util/model
! modules: once with the reference's molint.F90 (oracle/_ref/dropin_ref) and once
! with mc_water_ls_mw_amd/fortran/energy_hip.F90 + libmw_hip.so
! (oracle/_ref/dropin_hip).  tests/test_gpu_fortran_dropin.py runs both on the same
! input and compares every printed energy.
program dropin_driver
  use constants,  only : dp
  use userparams, only : nwater, num_lattices, model_type
  use model,      only : create_model, hmatrix, ljr, volume
  use energy
  implicit none
  character(len=512) :: path
  integer :: n, nlat, ils, i, k, nmoves, imol, refresh, u
  integer(kind=8) :: state
  real(kind=dp) :: d(3), old(2), new(2), backup(2), scale, xi
  real(kind=dp), allocatable :: keep(:,:)
  interface
     ! oracle/ref_scrub.c: zero the stack below this frame before every local-energy call, so the
     ! reference build is deterministic despite its uninitialised scratch array (SURVEY.md G2);
     ! harmless for the HIP build
     subroutine mw_scrub_stack() bind(C, name="mw_scrub_stack")
     end subroutine mw_scrub_stack
  end interface

  call get_command_argument(1, path)
  u = 20
  open(unit=u, file=trim(path), status='old')
  read(u,*) n, nlat, nmoves, refresh, state
  nwater = n
  num_lattices = nlat
  model_type = "mW"
  call create_model()
  do ils = 1, nlat
     read(u,*) hmatrix(:,:,ils)
     do i = 1, n
        read(u,*) ljr(:,1,i,ils)
     end do
  end do
  close(u)

  call energy_init()
  do ils = 1, nlat
     write(*,'(A,I2,ES26.17E3)') 'init  ', ils, model_energy(ils)
     write(*,'(A,I2,I6)') 'nivec ', ils, nivect(ils)
  end do

  do k = 1, nmoves
     if (mod(k, refresh) == 0) then                       ! mc_moves.F90:218-222
        do ils = 1, nlat
           call compute_neighbours(ils)
        end do
     end if
     imol = min(int(lcg()*real(n,kind=dp)) + 1, n)        ! mc_moves.F90:1001-1002
     do ils = 1, nlat
        call mw_scrub_stack()
        old(ils) = compute_local_real_energy(imol, ils)   ! :1010
        backup(ils) = model_energy(ils)                   ! :1013
        model_energy(ils) = model_energy(ils) - old(ils)  ! :1016
     end do
     d(1) = (2.0_dp*lcg() - 1.0_dp)*1.2_dp
     d(2) = (2.0_dp*lcg() - 1.0_dp)*1.2_dp
     d(3) = (2.0_dp*lcg() - 1.0_dp)*1.2_dp
     do ils = 1, nlat
        ljr(:,1,imol,ils) = ljr(:,1,imol,ils) + d(:)      ! :1079
        call mw_scrub_stack()
        new(ils) = compute_local_real_energy(imol, ils)   ! :1083
        model_energy(ils) = model_energy(ils) + new(ils)  ! :1087
     end do
     xi = lcg()
     if (xi < 0.5_dp) then                                ! reject: silent revert, :1182-1192
        do ils = 1, nlat
           ljr(:,1,imol,ils) = ljr(:,1,imol,ils) - d(:)
           model_energy(ils) = backup(ils)
        end do
     end if
     do ils = 1, nlat
        write(*,'(A,I2,I7,2ES26.17E3)') 'move  ', ils, imol, old(ils), new(ils)
     end do
  end do

  do ils = 1, nlat                                        ! drift check, mc_moves.F90:1786-1792
     write(*,'(A,I2,ES26.17E3)') 'accum ', ils, model_energy(ils)
     call compute_model_energy(ils)
     write(*,'(A,I2,ES26.17E3)') 'fresh ', ils, model_energy(ils)
  end do

  allocate(keep(3,n))
  scale = 1.013_dp
  do ils = 1, nlat                                        ! volume move, mc_moves.F90:1269-1357
     keep(:,:) = ljr(:,1,:,ils)
     hmatrix(:,:,ils) = hmatrix(:,:,ils)*scale
     ljr(:,1,:,ils) = ljr(:,1,:,ils)*scale
     call compute_ivects(ils)
     call compute_model_energy(ils)
     write(*,'(A,I2,ES26.17E3)') 'vol+  ', ils, model_energy(ils)
     hmatrix(:,:,ils) = hmatrix(:,:,ils)/scale            ! rejected: restore, :1410-1530
     ljr(:,1,:,ils) = keep(:,:)
     call compute_ivects(ils)                             ! ... and NOTHING else (mc_moves.F90:1510-1514):
     do i = 1, 5                                          ! the next translation moves see the restored box
        call mw_scrub_stack()
        write(*,'(A,I2,ES26.17E3)') 'lrej  ', ils, compute_local_real_energy(1 + mod(7*i, n), ils)
     end do
     call compute_model_energy(ils)
     write(*,'(A,I2,ES26.17E3)') 'vol0  ', ils, model_energy(ils)
     call compute_neighbours(ils)
     call compute_model_energy(ils)
     write(*,'(A,I2,ES26.17E3)') 'relst ', ils, model_energy(ils)
     write(*,'(A,I2,ES26.17E3)') 'volum ', ils, volume(ils)
  end do

  call energy_deinit()

contains

  real(kind=dp) function lcg()
    ! 48-bit linear congruential generator (drand48 constants); identical in both builds
    state = iand(state*25214903917_8 + 11_8, 281474976710655_8)
    lcg = real(state, kind=dp)/281474976710656.0_dp
  end function lcg

end program dropin_driver
