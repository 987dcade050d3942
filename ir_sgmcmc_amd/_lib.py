"""ctypes binding of include/irsgmcmc.h (the C-ABI boundary of the HIP library).

The library is loaded lazily on first use and its absence is a hard error: the product has NO CPU fallback.
Tensors cross the boundary as raw device pointers (`tensor.data_ptr()`) plus the current HIP stream; PyTorch-ROCm
is only the allocator / stream provider.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('IRS_LIB') or os.path.join(_HERE, 'csrc', 'libirsgmcmc.so')  # IRS_LIB: tuning builds

IRS_MAX_COMPONENTS = 8
IRS_MAX_CHAINS = 8
IRS_MAX_HALF_WIDTH = 4
IRS_DATA_GMM_LCC, IRS_DATA_SSD = 0, 1
IRS_REG_L2, IRS_REG_LOGNORMAL, IRS_REG_STUDENT, IRS_REG_LOGNORMAL_L2 = 0, 1, 2, 3


class IrsConfig(C.Structure):
    _fields_ = [
        ('dims', C.c_int32 * 3), ('cps', C.c_int32 * 3), ('no_chains', C.c_int32), ('no_steps', C.c_int32),
        ('sobolev_s', C.c_int32), ('sobolev_kernel', C.c_float * (2 * IRS_MAX_HALF_WIDTH + 1)),
        ('lr', C.c_float), ('uniform_alpha', C.c_float), ('virtual_decimation', C.c_int32),
        ('data_loss', C.c_int32), ('lcc_s', C.c_int32), ('gmm_components', C.c_int32), ('ssd_sigma', C.c_float),
        ('gmm_lr_log_std', C.c_float), ('gmm_lr_logits', C.c_float), ('gmm_lr_decay', C.c_float),
        ('adam_beta1', C.c_float), ('adam_beta2', C.c_float), ('adam_eps', C.c_float),
        ('scale_prior_loc', C.c_float), ('scale_prior_scale', C.c_float),
        ('dirichlet_concentration', C.c_float * IRS_MAX_COMPONENTS),
        ('reg_loss', C.c_int32), ('reg_learnable', C.c_int32), ('w_reg', C.c_float), ('dof', C.c_double),
        ('reg_lr0', C.c_float), ('reg_lr1', C.c_float), ('reg_lr_decay', C.c_float),
        ('loc_prior_nu', C.c_float), ('loc_prior_w_reg', C.c_float),
        ('reg_scale_prior_loc', C.c_float), ('reg_scale_prior_scale', C.c_float),
        ('w_reg_prior_shape', C.c_double), ('w_reg_prior_rate', C.c_double),
        ('seed', C.c_uint64),
    ]


class IrsState(C.Structure):
    _fields_ = [
        ('gmm_log_std', C.c_float * IRS_MAX_COMPONENTS), ('gmm_logits', C.c_float * IRS_MAX_COMPONENTS),
        ('gmm_adam_m', (C.c_double * IRS_MAX_COMPONENTS) * 2), ('gmm_adam_v', (C.c_double * IRS_MAX_COMPONENTS) * 2),
        ('gmm_adam_step', C.c_int64 * 2),
        ('reg_param', C.c_double * 2), ('reg_adam_m', C.c_double * 2), ('reg_adam_v', C.c_double * 2),
        ('reg_adam_step', C.c_int64 * 2),
        ('iteration', C.c_uint64),
    ]


class IrsScalars(C.Structure):
    _fields_ = [(n, C.c_double * IRS_MAX_CHAINS) for n in ('alpha', 'data_term', 'reg_term', 'reg_energy', 'n_mask')]


class IrsIO(C.Structure):
    _fields_ = [
        ('fixed_im', C.c_void_p), ('moving_im', C.c_void_p), ('mask', C.c_void_p),
        ('fixed_chains', C.c_int32), ('moving_chains', C.c_int32), ('mask_chains', C.c_int32),
        ('v', C.c_void_p), ('sigma', C.c_void_p), ('eps', C.c_void_p), ('unif', C.c_void_p),
        ('curr_state', C.c_void_p), ('im_moving_warped', C.c_void_p), ('residuals', C.c_void_p),
        ('displacement', C.c_void_p), ('transformation', C.c_void_p), ('grad_v', C.c_void_p),
    ]


class IrsTimings(C.Structure):
    _fields_ = [(n, C.c_float) for n in ('total_ms', 'exp_fwd_ms', 'exp_bwd_kernel_ms', 'exp_bwd_total_ms', 'smooth_ms',
                                         'data_ms', 'update_ms', 'exp_bwd_primary_avg_ms')]


class IrsXfer(C.Structure):
    _fields_ = [('ptr', C.c_void_p), ('bytes', C.c_size_t), ('peer', C.c_int32), ('recv', C.c_int32)]


class IrsSlabConfig(C.Structure):
    _fields_ = [('ghost_max', C.c_int32), ('margin', C.c_int32)]


class IrsSlabLayout(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('rank', 'world', 'a', 'b', 'lo', 'hi', 'margin', 'ghost_max')]


class IrsSlabStatus(C.Structure):
    _fields_ = [('transitions', C.c_uint64), ('exact_transitions', C.c_uint64), ('exchanges', C.c_uint64),
                ('exchanged_bytes', C.c_uint64), ('mispredictions', C.c_uint64), ('last_fwd_rounds', C.c_int32),
                ('last_bwd_rounds', C.c_int32)]


class IrsSlabTimelineEntry(C.Structure):
    _fields_ = [('kind', C.c_int32), ('stage', C.c_int32), ('k', C.c_int32), ('width', C.c_int32), ('ready_us', C.c_float),
                ('handover_us', C.c_float), ('wait_at_us', C.c_float), ('stall_us', C.c_float)]


class IrsSlabOp(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('kind', 'stage', 'k', 'lo0', 'hi0', 'lo1', 'hi1', 'in0', 'in1', 'reach', 'out', 'width', 'id')]


IRS_OP_LAUNCH, IRS_OP_EXCHANGE, IRS_OP_ALLREDUCE, IRS_OP_WAIT = range(4)
(IRS_SG_PERTURB, IRS_SG_COPY_V, IRS_SG_SMOOTH, IRS_SG_ENERGY, IRS_SG_EXP_FWD, IRS_SG_OUTPUTS, IRS_SG_WARP, IRS_SG_RESIDUAL, IRS_SG_STATS,
 IRS_SG_DATA_BWD, IRS_SG_WARP_BWD, IRS_SG_EXP_BWD, IRS_SG_UPDATE, IRS_SG_FFD_UP, IRS_SG_FFD_ADJ) = range(15)
IRS_SG_CHAIN_SCALAR, IRS_SG_REG_SCALAR, IRS_SG_FINALIZE = 32, 33, 34
(IRS_SB_V, IRS_SB_NOISY, IRS_SB_VS, IRS_SB_WARPED, IRS_SB_Z, IRS_SB_GM, IRS_SB_GRAD_A, IRS_SB_GRAD_B, IRS_SB_DENSE, IRS_SB_GRAD_C) = range(10)
IRS_SB_STEP0 = 16
IRS_COMM_ID_BYTES = 128
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(IrsXfer), C.c_int, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)

_P, _I, _F, _U64 = C.c_void_p, C.c_int, C.c_float, C.c_uint64
_I32P = C.POINTER(C.c_int32)

# name -> argtypes (restype is int unless listed in _RESTYPES); must match include/irsgmcmc.h exactly
SIGNATURES = {
    'irs_perturb_smooth': [_P, _P, _P, _F, _P, _I, _I, _I, _I, _I, _P, _P, _U64, _U64, _P],
    'irs_svf_exp_fwd': [_P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'irs_svf_exp_bwd': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'irs_ffd_up': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'irs_ffd_adjoint': [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _P],
    'irs_warp_fwd': [_P, _I, _P, _P, _F, _P, _I, _I, _I, _I, _U64, _U64, _P],
    'irs_warp_bwd': [_P, _I, _P, _P, _F, _P, _P, _I, _I, _I, _I, _U64, _U64, _P],
    'irs_warp_transformation': [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    'irs_warp_nearest_u8': [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    'irs_warp_nearest_i16': [_P, _I, _P, _P, _I, _I, _I, _I, _P],
    'irs_lcc_normalise': [_P, _P, _P, _I, _I, _I, _I, _I, _P],
    'irs_lcc_map_fwd': [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'irs_lcc_map_bwd': [_P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'irs_reg_energy': [_P, _P, _P, _I, _I, _I, _I, _P],
    'irs_reduce_scratch_doubles': [],
    'irs_gradient_operator': [_P, _P, _I, _I, _I, _I, _I, _P],
    'irs_log_det_jacobian': [_P, _P, _P, _I, _I, _I, _I, _P],
    'irs_create': [C.POINTER(IrsConfig), C.POINTER(_P)],
    'irs_destroy': [_P],
    'irs_workspace_bytes': [_P],
    'irs_velocity_dims': [_P, C.POINTER(C.c_int32 * 3)],
    'irs_set_fixed': [_P, _P, _I, _P],
    'irs_get_state': [_P, C.POINTER(IrsState), _P],
    'irs_set_state': [_P, C.POINTER(IrsState), _P],
    'irs_get_scalars': [_P, C.POINTER(IrsScalars), _P],
    'irs_gmm_init': [_P, C.POINTER(IrsIO), _P, _I, _P],
    'irs_transition': [_P, C.POINTER(IrsIO), _P],
    'irs_flush': [_P, _P],
    'irs_recovered_transitions': [_P, C.POINTER(C.c_uint64)],
    'irs_transition_timed': [_P, C.POINTER(IrsIO), _P, C.POINTER(IrsTimings)],
    'irs_comm_unique_id': [C.POINTER(C.c_uint8 * IRS_COMM_ID_BYTES)],
    'irs_comm_create_rccl': [C.POINTER(C.c_uint8 * IRS_COMM_ID_BYTES), _I, _I, C.POINTER(_P)],
    'irs_comm_create_ipc': [C.c_char_p, _I, _I, C.POINTER(_P)],
    'irs_comm_create_callbacks': [EXCHANGE_FN, ALLREDUCE_FN, _P, _I, _I, C.POINTER(_P)],
    'irs_comm_destroy': [_P],
    'irs_comm_probe': [_P, C.c_size_t, C.c_size_t, _I, _P, C.POINTER(C.c_double * 2)],
    'irs_comm_describe': [_P, C.c_char_p, C.c_size_t],
    'irs_comm_rank': [_P],
    'irs_comm_world': [_P],
    'irs_comm_selftest': [_P, _P],
    'irs_slab_plan_layout': [C.POINTER(IrsConfig), C.POINTER(IrsSlabConfig), _I, _I, C.POINTER(IrsSlabLayout)],
    'irs_slab_create': [C.POINTER(IrsConfig), C.POINTER(IrsSlabConfig), _P, C.POINTER(_P)],
    'irs_slab_get_layout': [_P, C.POINTER(IrsSlabLayout)],
    'irs_slab_transition': [_P, C.POINTER(IrsIO), _P],
    'irs_slab_gmm_init': [_P, C.POINTER(IrsIO), _P, _I, _P],
    'irs_slab_status_get': [_P, C.POINTER(IrsSlabStatus), _P],
    'irs_slab_timeline_arm': [_P, _I],
    'irs_slab_timeline_get': [_P, C.POINTER(IrsSlabTimelineEntry), _I, _I32P, C.POINTER(C.c_float), _P],
    'irs_slab_trace': [C.POINTER(IrsConfig), C.POINTER(IrsSlabConfig), _I, _I, _I32P, C.POINTER(IrsSlabOp), _I, _I32P],
    'irs_slab_plan_rounds': [_I32P, _I, _I, _I, _I, _I32P, _I32P, _I32P, _I32P, _I32P, _I32P],
    'irs_option_set': [_P, C.c_char_p, _I],
    'irs_last_error': [],
    'irs_version': [],
}
_RESTYPES = {'irs_reduce_scratch_doubles': C.c_size_t, 'irs_workspace_bytes': C.c_size_t, 'irs_destroy': None,
             'irs_comm_destroy': None,
             'irs_last_error': C.c_char_p, 'irs_version': C.c_char_p}

_lib = None


class IrsError(RuntimeError):
    pass


def load(path=None):
    """dlopen the C-ABI library and bind every symbol of the header.  Raises if the library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.isfile(p):
        raise IrsError(f'{p} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                       f'(or `make -C ir_sgmcmc_amd/csrc`). There is no CPU fallback.')
    lib = C.CDLL(p)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export what the header declares
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, C.c_int)
    if path is None:
        _lib = lib
    return lib


def option_set(name, value, ctx=None):
    """irs_option_set: a tuning / test switch by name, on one context or (ctx None) process-wide"""
    check(load().irs_option_set(ctx, name.encode(), int(value)))


def check(rc):
    if rc != 0:
        raise IrsError(load().irs_last_error().decode())


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev_ptr(t, dtype=None, allow_none=False):
    """Raw device pointer of a contiguous CUDA(ROCm) tensor; refuses host tensors (no CPU path in the product)."""
    if t is None:
        if allow_none:
            return None
        raise IrsError('required tensor is None')
    if not t.is_cuda:
        raise IrsError('ir_sgmcmc_amd operators run on the GPU only (got a CPU tensor); there is no CPU fallback')
    if dtype is not None and t.dtype != dtype:
        raise IrsError(f'expected {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise IrsError('tensor must be contiguous')
    return C.c_void_p(t.data_ptr())
