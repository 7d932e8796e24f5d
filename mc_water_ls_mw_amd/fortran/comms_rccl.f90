!=============================================================================!
!                     C O M M S   (RCCL over xGMI)                            !
!=============================================================================!
! Drop-in replacement for the reference's `module comms` (comms_mpi.f90 /       !
! comms_serial.f90): same module name, same public routines and argument        !
! lists, same public variables (myrank, size, *_last_sync), so main.f90,        !
! io.f90, init.f90 and mc_moves.F90 compile and link against it unchanged.      !
! One process per GPU; every collective goes through libmw_comms.so             !
! (include/mw_comms.h) to RCCL -- no MPI library, no mpif.h.  Rank, size, GPU   !
! and the bootstrap file come from the environment (see mw_comms.h): start the  !
! ranks with any launcher that sets RANK / WORLD_SIZE / LOCAL_RANK, e.g.        !
!   python -m torch.distributed.run --no-python --nproc-per-node 8 ./mc_water   !
!                                                                               !
! What the tables exchange, exactly as comms_mpi.f90:244-277,461-530 has it:    !
!   table <- allreduce_sum(table - last_sync) + last_sync ; last_sync <- table  !
! The window joins ('dd', comms_mpi.f90:299-459) gather every rank's window to  !
! EVERY rank (one all-gather) and stitch them there in rank order -- the same    !
! arithmetic the reference runs on rank 0 before broadcasting the result.        !
!=============================================================================!
module comms

  use iso_c_binding
  use constants, only : dp

  implicit none

  ! kept for source compatibility (comms_mpi.f90:16-19); the parallel-tempering buffers are never used
  real(kind=dp),allocatable,dimension(:),save :: Rbuffer,Sbuffer
  real(kind=dp),allocatable,dimension(:),save :: hist_last_sync,eta_last_sync
  real(kind=dp),allocatable,dimension(:),save :: uhist_last_sync

  integer :: myrank,size

  interface
     function mw_comms_init(rank_out,size_out) bind(C,name='mw_comms_init') result(rc)
       import :: c_int
       integer(c_int),intent(out) :: rank_out,size_out
       integer(c_int) :: rc
     end function mw_comms_init
     function mw_comms_allreduce_sum(buf,n) bind(C,name='mw_comms_allreduce_sum') result(rc)
       import :: c_int,c_double
       real(c_double),intent(inout) :: buf(*)
       integer(c_int),value :: n
       integer(c_int) :: rc
     end function mw_comms_allreduce_sum
     function mw_comms_allreduce_max(buf,n) bind(C,name='mw_comms_allreduce_max') result(rc)
       import :: c_int,c_double
       real(c_double),intent(inout) :: buf(*)
       integer(c_int),value :: n
       integer(c_int) :: rc
     end function mw_comms_allreduce_max
     function mw_comms_allgather(mine,all,n) bind(C,name='mw_comms_allgather') result(rc)
       import :: c_int,c_double
       real(c_double),intent(in) :: mine(*)
       real(c_double),intent(out) :: all(*)
       integer(c_int),value :: n
       integer(c_int) :: rc
     end function mw_comms_allgather
     function mw_comms_bcast(buf,nbytes,root) bind(C,name='mw_comms_bcast') result(rc)
       import :: c_int,c_long,c_ptr
       type(c_ptr),value :: buf
       integer(c_long),value :: nbytes
       integer(c_int),value :: root
       integer(c_int) :: rc
     end function mw_comms_bcast
     function mw_comms_sendrecv(buf,nbytes,snode,rnode) bind(C,name='mw_comms_sendrecv') result(rc)
       import :: c_int,c_long,c_ptr
       type(c_ptr),value :: buf
       integer(c_long),value :: nbytes
       integer(c_int),value :: snode,rnode
       integer(c_int) :: rc
     end function mw_comms_sendrecv
     function mw_comms_barrier() bind(C,name='mw_comms_barrier') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function mw_comms_barrier
     function mw_comms_finalize() bind(C,name='mw_comms_finalize') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function mw_comms_finalize
     function mw_comms_abort() bind(C,name='mw_comms_abort') result(rc)
       import :: c_int
       integer(c_int) :: rc
     end function mw_comms_abort
     function mw_comms_last_error() bind(C,name='mw_comms_last_error') result(p)
       import :: c_ptr
       type(c_ptr) :: p
     end function mw_comms_last_error
  end interface

contains

  subroutine comms_check(rc,what)
    ! Stops the run with the library's message, as the reference stops on an MPI error.
    integer(c_int),intent(in) :: rc
    character(len=*),intent(in) :: what
    character(kind=c_char),pointer :: msg(:)
    integer :: k
    if (rc==0) return
    call c_f_pointer(mw_comms_last_error(),msg,(/512/))
    write(0,'("Error in ",A," : ")',advance='no')what
    do k = 1,512
       if (msg(k)==c_null_char) exit
       write(0,'(A1)',advance='no')msg(k)
    end do
    write(0,*)
    ! Nothing of this rank is left for the next job to trip over (its id file), and nothing here waits for a peer that may be
    ! gone: the communicator is aborted, not destroyed (a stream synchronisation or ncclCommDestroy could block for ever).
    k = mw_comms_abort()
    stop 'comms (RCCL) failure'
  end subroutine comms_check

  subroutine Comms_Initialise()
    ! comms_mpi.f90:26-71
    integer(c_int) :: r,s
    call comms_check(mw_comms_init(r,s),'comms_initialise')
    myrank = r
    size   = s
  end subroutine Comms_Initialise

  subroutine comms_allocate()
    ! comms_mpi.f90:73-104 (the last-synchronised tables of the delta scheme)
    use userparams, only : nbins,samplerun
    integer :: ierr
    allocate(eta_last_sync(1:nbins),stat=ierr)
    if (ierr==0) allocate(hist_last_sync(1:nbins),stat=ierr)
    if (ierr==0 .and. samplerun) allocate(uhist_last_sync(1:nbins),stat=ierr)
    if (ierr/=0) stop 'Error allocating bin sync arrays'
    eta_last_sync  = 0.0_dp
    hist_last_sync = 0.0_dp
    if (samplerun) uhist_last_sync = 0.0_dp
  end subroutine comms_allocate

  !---------------------------------------------------------------------------!
  ! Broadcasts from rank 0 (comms_mpi.f90:107-166,224-242).  The actual        !
  ! argument is the first element of `Length` contiguous values.               !
  !---------------------------------------------------------------------------!
  subroutine Comms_BcastReal(Rvalue,Length)
    real(kind=dp),intent(inout),target :: Rvalue
    integer,intent(in) :: Length
    call comms_check(mw_comms_bcast(c_loc(Rvalue),int(Length,c_long)*c_sizeof(Rvalue),0_c_int),'comms_bcastreal')
  end subroutine Comms_BcastReal

  subroutine Comms_BcastInt(Ivalue,Length)
    integer,intent(inout),target :: Ivalue
    integer,intent(in) :: Length
    call comms_check(mw_comms_bcast(c_loc(Ivalue),int(Length,c_long)*int(storage_size(Ivalue)/8,c_long),0_c_int),'comms_bcastint')
  end subroutine Comms_BcastInt

  subroutine Comms_BcastLog(Lvalue,Length)
    logical,intent(inout),target :: Lvalue
    integer,intent(in) :: Length
    call comms_check(mw_comms_bcast(c_loc(Lvalue),int(Length,c_long)*int(storage_size(Lvalue)/8,c_long),0_c_int),'comms_bcastlog')
  end subroutine Comms_BcastLog

  subroutine Comms_BcastChar(Cvalue,Length)
    character,intent(inout),target :: Cvalue
    integer,intent(in) :: Length
    call comms_check(mw_comms_bcast(c_loc(Cvalue),int(Length,c_long),0_c_int),'comms_bcastchar')
  end subroutine Comms_BcastChar

  !---------------------------------------------------------------------------!
  ! Point to point (comms_mpi.f90:168-222): snode's values arrive at rnode.    !
  !---------------------------------------------------------------------------!
  subroutine comms_p2preal(Rvalue,length,snode,rnode)
    real(kind=dp),intent(inout),target :: Rvalue
    integer,intent(in) :: length,snode,rnode
    call comms_check(mw_comms_sendrecv(c_loc(Rvalue),int(length,c_long)*c_sizeof(Rvalue),int(snode,c_int),int(rnode,c_int)),'comms_p2preal')
  end subroutine comms_p2preal

  subroutine comms_p2pint(Ivalue,length,snode,rnode)
    integer,intent(inout),target :: Ivalue
    integer,intent(in) :: length,snode,rnode
    call comms_check(mw_comms_sendrecv(c_loc(Ivalue),int(length,c_long)*int(storage_size(Ivalue)/8,c_long), &
         int(snode,c_int),int(rnode,c_int)),'comms_p2pint')
  end subroutine comms_p2pint

  !---------------------------------------------------------------------------!
  ! The delta all-reduces of the multicanonical tables.                        !
  !---------------------------------------------------------------------------!
  subroutine delta_allreduce(table,last,Length,what)
    integer,intent(in) :: Length
    real(kind=dp),intent(inout),dimension(1:Length) :: table,last
    character(len=*),intent(in) :: what
    real(kind=dp),dimension(1:Length) :: buff
    buff = table - last                                        ! what this rank added since the last sync
    call comms_check(mw_comms_allreduce_sum(buff,int(Length,c_int)),what)
    table = buff + last
    last  = table
  end subroutine delta_allreduce

  subroutine comms_allreduce_eta(weight,Length)
    ! comms_mpi.f90:244-277
    integer,intent(in) :: Length
    real(kind=dp),intent(inout),dimension(1:Length) :: weight
    call delta_allreduce(weight,eta_last_sync,Length,'comms_allreduce_eta')
  end subroutine comms_allreduce_eta

  subroutine comms_allreduce_hist(histogram,Length)
    ! comms_mpi.f90:461-494 (the reference holds a barrier first; the collective itself synchronises)
    integer,intent(in) :: Length
    real(kind=dp),intent(inout),dimension(Length) :: histogram
    call delta_allreduce(histogram,hist_last_sync,Length,'comms_allreduce_hist')
  end subroutine comms_allreduce_hist

  subroutine comms_allreduce_uhist(histogram,Length)
    ! comms_mpi.f90:496-530
    integer,intent(in) :: Length
    real(kind=dp),intent(inout),dimension(Length) :: histogram
    call delta_allreduce(histogram,uhist_last_sync,Length,'comms_allreduce_uhist')
  end subroutine comms_allreduce_uhist

  subroutine comms_set_histogram(hist_in,length)
    ! comms_mpi.f90:533-548
    integer,intent(in) :: length
    real(kind=dp),dimension(length),intent(in) :: hist_in
    hist_last_sync = hist_in
  end subroutine comms_set_histogram

  subroutine comms_set_uhistogram(hist_in,length)
    ! comms_mpi.f90:550-566
    integer,intent(in) :: length
    real(kind=dp),dimension(length),intent(in) :: hist_in
    uhist_last_sync = hist_in
  end subroutine comms_set_uhistogram

  subroutine comms_get_max(myval,maxval)
    ! comms_mpi.f90:279-297
    real(kind=dp),intent(in)  :: myval
    real(kind=dp),intent(out) :: maxval
    real(kind=dp),dimension(1) :: v
    v(1) = myval
    call comms_check(mw_comms_allreduce_max(v,1_c_int),'comms_get_max')
    maxval = v(1)
  end subroutine comms_get_max

  !---------------------------------------------------------------------------!
  ! Window joins of the 'dd' strategy.  Rank r contributes the bins above      !
  ! r*bins_per_window, shifted so that its mean over the 2*overlap+1 bins      !
  ! around that seam agrees with what has been joined so far.                  !
  !---------------------------------------------------------------------------!
  subroutine comms_join_eta(weight,Length,overlap,joined)
    ! comms_mpi.f90:381-459
    integer,intent(in) :: length,overlap
    real(kind=dp),intent(in),dimension(1:length) :: weight
    real(kind=dp),intent(out),dimension(1:length) :: joined
    real(kind=dp),allocatable,dimension(:,:) :: windows
    real(kind=dp) :: shift,myave,nextav,mid
    integer :: bins_per_window,irank,k,seam

    allocate(windows(1:length,0:size-1))
    call comms_check(mw_comms_allgather(weight,windows,int(length,c_int)),'comms_join_eta')
    bins_per_window = length/size
    joined = windows(:,0)
    do irank = 1,size-1
       seam = irank*bins_per_window
       myave  = 0.0_dp
       nextav = 0.0_dp
       do k = seam-overlap,seam+overlap
          myave  = myave  + joined(k)
          nextav = nextav + windows(k,irank)
       end do
       shift = myave/real(2*overlap+1,kind=dp) - nextav/real(2*overlap+1,kind=dp)
       do k = seam+1,length
          joined(k) = windows(k,irank) + shift
       end do
    end do
    mid = joined(length/2+1)
    do k = 1,length
       joined(k) = joined(k) - mid
    end do
    deallocate(windows)
  end subroutine comms_join_eta

  subroutine comms_join_uhist(uhist,length,overlap,joined)
    ! comms_mpi.f90:299-379: the same in log space, no shift of the middle bin
    integer,intent(in) :: length,overlap
    real(kind=dp),intent(in),dimension(1:length) :: uhist
    real(kind=dp),intent(out),dimension(1:length) :: joined
    real(kind=dp),allocatable,dimension(:,:) :: windows
    real(kind=dp) :: shift,myave,nextav
    integer :: bins_per_window,irank,k,seam

    allocate(windows(1:length,0:size-1))
    call comms_check(mw_comms_allgather(uhist,windows,int(length,c_int)),'comms_join_uhist')
    bins_per_window = length/size
    joined = windows(:,0)
    do irank = 1,size-1
       seam = irank*bins_per_window
       myave  = 0.0_dp
       nextav = 0.0_dp
       do k = seam-overlap,seam+overlap
          myave  = myave  + log(joined(k))
          nextav = nextav + log(windows(k,irank))
       end do
       shift = myave/real(2*overlap+1,kind=dp) - nextav/real(2*overlap+1,kind=dp)
       if (shift/=shift) shift = 0.0_dp                        ! NaN from empty seam bins
       do k = seam+1,length
          joined(k) = windows(k,irank)*exp(shift)
       end do
    end do
    deallocate(windows)
  end subroutine comms_join_uhist

  subroutine Comms_Finalise()
    ! comms_mpi.f90:569-599
    if (allocated(eta_last_sync))   deallocate(eta_last_sync)
    if (allocated(hist_last_sync))  deallocate(hist_last_sync)
    if (allocated(uhist_last_sync)) deallocate(uhist_last_sync)
    call comms_check(mw_comms_finalize(),'comms_finalise')
  end subroutine Comms_Finalise

  subroutine comms_barrier()
    ! comms_mpi.f90:601-618
    call comms_check(mw_comms_barrier(),'comms_barrier')
  end subroutine comms_barrier

end module comms
