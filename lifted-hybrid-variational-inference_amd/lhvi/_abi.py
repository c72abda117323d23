"""ctypes binding of ``liblhvi.so`` (the C ABI declared in ``include/lhvi.h``).

PyTorch-ROCm is used only as the device allocator / stream provider: tensors are created with torch,
their ``data_ptr()`` goes into the C structs, and launches run on ``torch.cuda.current_stream()``.
There is no CPU fallback: if the library is missing or no GPU is visible, calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('LHVI_LIB') or os.path.join(os.path.dirname(_HERE), 'csrc', 'liblhvi.so')

c_i32p = C.c_void_p
c_f64p = C.c_void_p


class LhviError(RuntimeError):
    pass


class GraphStruct(C.Structure):
    _fields_ = [
        ('V', C.c_int32), ('F', C.c_int32), ('E', C.c_int32), ('nnz', C.c_int32),
        ('fac_ptr', C.c_void_p), ('edge_var', C.c_void_p), ('edge_fac', C.c_void_p), ('edge_canon', C.c_void_p),
        ('var_ptr', C.c_void_p), ('var_edge', C.c_void_p), ('edge_count', C.c_void_p), ('fac_pot', C.c_void_p),
        ('var_value', C.c_void_p), ('var_dom', C.c_void_p), ('var_mult', C.c_void_p), ('fac_mult', C.c_void_p),
        ('D', C.c_int32),
        ('dom_cont', C.c_void_p), ('dom_lo', C.c_void_p), ('dom_hi', C.c_void_p), ('dom_ptr', C.c_void_p),
        ('dom_val', C.c_void_p), ('edge_value', C.c_void_p), ('slot_var', C.c_void_p),
        ('hub_vars', C.c_void_p), ('n_hubs', C.c_int32),
    ]


class GabpPlanStruct(C.Structure):
    _fields_ = [('pslot', C.c_void_p), ('info', C.c_void_p), ('count', C.c_void_p), ('n_hub_rows', C.c_int32), ('rec', C.c_void_p), ('pot_words', C.c_void_p), ('seg', C.c_void_p), ('n_seg', C.c_int32)]


class PotsStruct(C.Structure):
    _fields_ = [('P', C.c_int32), ('kind', C.c_void_p), ('off', C.c_void_p), ('param', C.c_void_p), ('interpreted', C.c_int32)]


class PbpStruct(C.Structure):
    _fields_ = [
        ('n', C.c_int32), ('T', C.c_int32), ('flags', C.c_uint32),
        ('var_threshold', C.c_double), ('max_log_value', C.c_double),
        ('particles', C.c_void_p), ('old_particles', C.c_void_p), ('np', C.c_void_p), ('uniq', C.c_void_p),
        ('q', C.c_void_p),
        ('fast_edges', C.c_void_p), ('n_fast', C.c_int32), ('generic_edges', C.c_void_p), ('n_generic', C.c_int32),
        ('generic_pts_log2', C.c_int32), ('fast_desc', C.c_void_p), ('heavy_desc', C.c_void_p), ('n_heavy', C.c_int32), ('light_desc', C.c_void_p), ('n_light', C.c_int32),
        ('bslot', C.c_void_p), ('brow_ptr', C.c_void_p), ('brow_off', C.c_void_p), ('brow_peer', C.c_void_p),
        ('recv', C.c_void_p), ('rank', C.c_int32), ('var_degree', C.c_void_p),
        ('var_lo', C.c_int32), ('var_hi', C.c_int32),
        ('f2v_ticket', C.c_void_p),
        ('prop_desc', C.c_void_p), ('n_prop_desc', C.c_int32),
        ('pair_desc', C.c_void_p), ('n_pair', C.c_int32),
        ('cq_desc', C.c_void_p), ('n_cq', C.c_int32),
        ('v2f_wide', C.c_void_p), ('n_v2f_wide', C.c_int32), ('v2f_narrow', C.c_void_p), ('n_v2f_narrow', C.c_int32),
        ('v2f_hub', C.c_void_p), ('n_v2f_hub', C.c_int32),
        ('v2f_mid16', C.c_void_p), ('n_v2f_mid16', C.c_int32), ('v2f_mid32', C.c_void_p), ('n_v2f_mid32', C.c_int32),
        ('prop_hub', C.c_void_p), ('n_prop_hub', C.c_int32), ('prop_partial', C.c_void_p),
        ('resample_vars', C.c_void_p), ('n_resample_vars', C.c_int32),
        ('small16_desc', C.c_void_p), ('n_small16', C.c_int32), ('small32_desc', C.c_void_p), ('n_small32', C.c_int32),
        ('halo_off', C.c_void_p), ('halo_buf', C.c_void_p),
    ]


class ViStruct(C.Structure):
    _fields_ = [
        ('K', C.c_int32), ('T', C.c_int32), ('Dmax', C.c_int32), ('quirks', C.c_int32),
        ('gh_x', C.c_void_p), ('gh_w', C.c_void_p), ('w', C.c_void_p), ('eta_c', C.c_void_p), ('eta_d', C.c_void_p),
        ('obs_var', C.c_void_p), ('var_N', C.c_void_p),
        ('fac_list', C.c_void_p), ('n_cc', C.c_int32), ('n_grp3', C.c_int32), ('n_grp6', C.c_int32), ('n_rest3', C.c_int32),
        ('n_rest6', C.c_int32), ('n_tiny', C.c_int32), ('tiny_par_words', C.c_int32), ('edge_axis', C.c_void_p),
    ]


class ViOptStruct(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ('w_tau', 'w', 'eta_c', 'tau_d', 'eta_d', 'm_w', 's_w', 'm_c', 's_c', 'm_d', 's_d',
                                          'g_w', 'g_c', 'g_d', 'fe')] + \
               [(n, C.c_double) for n in ('lr', 'b1', 'b2', 'eps', 'var_min')] + [('t', C.c_int32)]


PBP_EP = 1
PBP_EPBP_DISCRETE = 2
PBP_SKIP_FAST = 4
PBP_SKIP_GENERIC = 8
PBP_SKIP_TERMS = 16
PBP_SKIP_HEAVY = 32
PBP_SKIP_LIGHT = 64
PBP_NO_GRID = 128
PBP_LEAVE_ROOM = 256
PBP_BOUNDARY_TOTALS = 2048
PBP_CQ = 512
PBP_NO_UNIQ = 4096
PBP_V2F_RECORDS = 8192
PBP_WIDE_PAIRS = 16384
PBP_SHARE_CUS = 32768
PBP_POW2_GROUPS = 65536
PBP_FUSED_RECORDS16 = 131072
PBP_SKIP_CQ = 1024
ABI_VERSION = 11            # LHVI_ABI_VERSION of include/lhvi.h (struct layouts)
PBP_DESC_BYTES = 128
COLOR_HASH, COLOR_SORT = 0, 1     # method of lhvi_color_refine_* (LHVI_COLOR_HASH / LHVI_COLOR_SORT)
HUB_DEGREE = 64              # LHVI_HUB_DEGREE
VI_GROUP_SLOTS, VI_GROUP_COMP, VI_TINY_NODES, VI_TINY_K, VI_TINY_PAR = 24, 48, 32, 2, 3072     # LHVI_VI_GROUP_SLOTS / LHVI_VI_GROUP_COMP

_G, _P, _S, _VI = C.POINTER(GraphStruct), C.POINTER(PotsStruct), C.POINTER(PbpStruct), C.POINTER(ViStruct)
_GP = C.POINTER(GabpPlanStruct)
_vp, _i32, _i64, _u32, _u64, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_double, C.c_size_t

# name -> (restype, argtypes); must list every symbol include/lhvi.h declares (tests/test_abi.py checks)
SIGNATURES = {
    'lhvi_version': (C.c_int, []),
    'lhvi_strerror': (C.c_char_p, [C.c_int]),
    'lhvi_last_hip_error': (C.c_int, []),
    'lhvi_device_count': (C.c_int, []),
    'lhvi_gabp_init': (C.c_int, [_G, _vp, _vp, _vp]),
    'lhvi_gabp_v2f': (C.c_int, [_G, _vp, _vp, _vp]),
    'lhvi_gabp_f2v': (C.c_int, [_G, _P, _vp, _vp, _vp]),
    'lhvi_gabp_run': (C.c_int, [_G, _P, _vp, _vp, C.c_int, _vp]),
    'lhvi_gabp_marginals': (C.c_int, [_G, _vp, _vp, _vp]),
    'lhvi_gabp_pull_workspace_bytes': (C.c_size_t, [_G]),
    'lhvi_gabp_pull': (C.c_int, [_G, _P, _GP, _vp, _vp, C.c_int, _vp]),
    'lhvi_gabp_run_pull': (C.c_int, [_G, _P, _GP, _vp, _vp, C.c_int, _vp, C.c_size_t, _vp]),
    'lhvi_gabp_graph_create': (C.c_int, [_G, _P, _GP, _vp, _vp, _vp, C.c_int, _vp, C.c_size_t, C.POINTER(C.c_void_p)]),
    'lhvi_gabp_graph_launch': (C.c_int, [_vp, _vp]),
    'lhvi_gabp_graph_destroy': (C.c_int, [_vp]),
    'lhvi_pbp_uniq': (C.c_int, [_G, _i32, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_classify': (C.c_int, [_G, _P, _S, _vp, _vp]),
    'lhvi_pbp_describe': (C.c_int, [_G, _P, _S, _vp, _i32, _vp, _vp]),
    'lhvi_pbp_describe_cq': (C.c_int, [_G, _P, _S, _vp, _i32, _vp, _vp]),
    'lhvi_log_likelihood_workspace_bytes': (C.c_size_t, [_G]),
    'lhvi_log_likelihood': (C.c_int, [_G, _P, _vp, _vp, _vp, C.c_size_t, _vp]),
    'lhvi_pbp_var_sum': (C.c_int, [_G, _S, _vp, _vp, _vp]),
    'lhvi_pbp_domain_grid': (C.c_int, [_G, _S, _vp, _vp]),
    'lhvi_pbp_refine_grid': (C.c_int, [_G, _S, _vp, _vp, _vp, _vp, _vp]),
    'lhvi_debug_exp': (C.c_int, [_vp, _vp, _i64, _vp]),
    'lhvi_debug_log': (C.c_int, [_vp, _vp, _i64, C.c_int32, _vp]),
    'lhvi_debug_exp_acc': (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    'lhvi_debug_exp_acc_floor': (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    'lhvi_pbp_v2f': (C.c_int, [_G, _S, _vp, _vp, _vp]),
    'lhvi_pbp_f2v': (C.c_int, [_G, _P, _S, _vp, _vp, _vp]),
    'lhvi_pbp_proposal': (C.c_int, [_G, _S, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_proposal_partial': (C.c_int, [_G, _S, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_proposal_finish': (C.c_int, [_G, _S, _vp, _vp, _vp]),
    'lhvi_pbp_boundary_pack': (C.c_int, [_G, _S, _vp, _vp, _i32, _vp, _vp, _vp]),
    'lhvi_pbp_boundary_reduce': (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_halo_pack': (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_halo_unpack': (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_init': (C.c_int, [_G, _S, _vp, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_resample': (C.c_int, [_G, _S, _vp, _u64, _u32, _vp, _vp]),
    'lhvi_pbp_edge_points': (C.c_int, [_G, _P, _S, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    'lhvi_pbp_resample_uniq': (C.c_int, [_G, _S, _vp, _u64, _u32, _vp, _vp, _vp]),
    'lhvi_pbp_belief_points': (C.c_int, [_G, _P, _S, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    'lhvi_pbp_map_brent': (C.c_int, [_G, _P, _S, _vp, _i64, _vp, _vp, _vp, _vp, _f64, _i32, _vp, _vp, _vp, _vp]),
    'lhvi_pbp_var_fused': (C.c_int, [_G, _S, _vp, _vp, _vp, _vp, _vp, _u64, _u32, _vp, _vp, _vp, _i32, _i32, _i32, _vp]),
    'lhvi_pbp_quad': (C.c_int, [_G, _P, _S, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp]),
    'lhvi_vi_workspace_bytes': (_sz, [_G, _VI]),
    'lhvi_vi_grad': (C.c_int, [_G, _P, _VI, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'lhvi_vi_adam_run': (C.c_int, [_G, _P, _VI, C.POINTER(ViOptStruct), _i32, _vp, _vp, _sz, _vp]),
    'lhvi_adam_step': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f64, _f64, _f64, _f64, _i32, _f64, _vp]),
    'lhvi_softmax_rows': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    'lhvi_color_workspace_bytes': (_sz, [_G]),
    'lhvi_color_refine_factors': (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    'lhvi_color_refine_rvs': (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _sz, _i32, _vp]),
    'lhvi_color_first_members': (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    'lhvi_color_segment_sums': (C.c_int, [_vp, _vp, _i32, _vp, _vp]),
}

_lib = None
MISSING = []


def lib():
    """Load liblhvi.so once; raise (never fall back) when it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise LhviError('HIP extension %s not built; run `python -c "import __graft_entry__ as g; g.build()"`'
                            % LIB_PATH)
        # PyTorch-ROCm ships its own copy of the HIP runtime: it has to be in the process BEFORE liblhvi.so is loaded, or the
        # library binds to the system's copy and its launches see none of torch's devices, streams or allocations
        # (hipErrorNoDevice on the first launch)
        _torch()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError:
                MISSING.append(name)        # build() and tests/test_abi.py require this list to be empty
                continue
            fn.restype, fn.argtypes = res, args
        if handle.lhvi_version() != ABI_VERSION:
            raise LhviError('liblhvi.so ABI version mismatch')
        _lib = handle
    return _lib


def check(rc):
    if rc != 0:
        l = lib()
        raise LhviError('liblhvi: %s (code %d, hipError %d)' % (l.lhvi_strerror(rc).decode(), rc, l.lhvi_last_hip_error()))


def _torch():
    import torch
    return torch


def require_gpu():
    torch = _torch()
    if not torch.cuda.is_available():
        raise LhviError('no MI355X/HIP device visible: the lhvi solvers have no CPU fallback '
                        '(the CPU restatement lives in oracle/ and is test infrastructure only)')
    return torch


def stream_ptr():
    return C.c_void_p(_torch().cuda.current_stream().cuda_stream)


def ptr(t):
    return C.c_void_p(0 if t is None else t.data_ptr())


def to_dev(a, device=None):
    torch = require_gpu()
    return torch.from_numpy(np.ascontiguousarray(a)).to(device or 'cuda', non_blocking=False)


def upload(arrays, device=None):
    """several host arrays to the device in ONE copy: they are packed (256-byte aligned) into a staging buffer, copied once, and
    returned as typed views of the device buffer (dict name -> tensor; a None entry stays None).  A lifted graph of a
    coarse-to-fine sweep is some twenty small arrays -- one transfer instead of twenty."""
    torch = require_gpu()
    items = [(k, np.ascontiguousarray(a)) for k, a in arrays.items() if a is not None]
    offs, total = [], 0
    for _, a in items:
        offs.append(total)
        total += (a.nbytes + 255) // 256 * 256
    if total > (1 << 28) or len(items) < 2:          # large graphs: plain per-array copies (no second host copy of 100s of MB)
        out = {k: to_dev(a, device) for k, a in items}
    else:
        stage = np.zeros(max(total, 256), dtype=np.uint8)
        for (k, a), o in zip(items, offs):
            stage[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
        dev = torch.from_numpy(stage).to(device or 'cuda', non_blocking=False)
        out = {}
        for (k, a), o in zip(items, offs):
            td = torch.from_numpy(np.zeros(0, dtype=a.dtype)).dtype
            out[k] = dev[o:o + a.nbytes].view(td).view(a.shape)
    for k, a in arrays.items():
        if a is None:
            out[k] = None
    return out


def device_potentials(flat):
    """(pot_off, pot_param, interpreted) as they go to the device.  The host table keeps every formula's bytecode (the CPU oracle
    interprets it); on the device a formula with a conditional-quadratic block is evaluated through the block alone
    (``cq_log_phi``, csrc/potential.hpp), so its row travels as ``[w, 0, 3, block]`` -- no program: the table is a fraction of the
    size (the variational kernels keep it in LDS).  `interpreted` = rows the device has to interpret (``lhvi_pots_t.interpreted``)."""
    def build():
        kind, off, par = flat.pot_kind, flat.pot_off, flat.pot_param
        rows, interpreted = [], 0
        for i in range(int(kind.size)):
            row = par[off[i]:off[i + 1]]
            if kind[i] == 8 and row.size > 2 and row[2] != 0:
                row = np.concatenate([[row[0], 0.0, 3.0], row[int(row[2]):]])
            elif kind[i] in (8, 9):
                interpreted += 1
            rows.append(row)
        new_off = np.zeros(kind.size + 1, dtype=np.int32)
        np.cumsum([r.size for r in rows], out=new_off[1:])
        new_par = np.concatenate(rows) if rows else np.zeros(0)
        return new_off, np.ascontiguousarray(new_par, dtype=np.float64), interpreted
    return flat._cached('device_potentials', (flat.pot_kind, flat.pot_off, flat.pot_param), build)


class DeviceGraph:
    """A ``FlatGraph`` resident in HBM plus the two C structs that point into it."""

    def __init__(self, flat, device=None):
        require_gpu()
        self.flat = flat
        host = {name: getattr(flat, name) for name in
                ('fac_ptr', 'edge_var', 'edge_fac', 'var_ptr', 'var_edge', 'fac_pot', 'var_value', 'var_dom',
                 'dom_cont', 'dom_lo', 'dom_hi', 'dom_ptr', 'dom_val', 'pot_kind')}
        host['pot_off'], host['pot_param'], interpreted = device_potentials(flat)
        self.pot_param_words = int(host['pot_param'].size)
        has_alias = flat._cached('has_alias', (flat.edge_canon,),
                                 lambda: bool((flat.edge_canon != np.arange(flat.E, dtype=np.int32)).any()))
        host['edge_canon'] = flat.edge_canon if has_alias else None
        host['edge_count'] = flat.edge_count if flat.lifted else None
        host['var_mult'] = flat.var_mult if flat.lifted else None
        host['fac_mult'] = flat.fac_mult if flat.lifted else None
        # denormalised copies for the Gaussian sweep (contiguous instead of gathered): built on the host for a small graph (one
        # upload), on the device for a large one (two gathers there instead of 120 MB more over PCIe at 10 M edges)
        large = flat.E >= (1 << 18)
        if not large:
            host['edge_value'] = np.ascontiguousarray(flat.var_value[flat.edge_var]) if flat.E else None
            host['slot_var'] = np.repeat(np.arange(flat.V, dtype=np.int32), np.diff(flat.var_ptr)) if flat.var_edge.size else None
        hubs = np.flatnonzero(np.diff(flat.var_ptr) > HUB_DEGREE).astype(np.int32)
        host['hub_vars'] = hubs if hubs.size else np.zeros(1, dtype=np.int32)   # non-NULL even when empty
        t = upload(host, device)
        if large:
            torch = _torch()
            t['edge_value'] = t['var_value'][t['edge_var'].long()]
            vp = t['var_ptr'].long()
            t['slot_var'] = torch.repeat_interleave(torch.arange(flat.V, dtype=torch.int32, device=vp.device), vp[1:] - vp[:-1],
                                                    output_size=int(flat.var_edge.size))
        self.t = t
        self.device = t['fac_ptr'].device
        g = GraphStruct()
        g.V, g.F, g.E, g.nnz = flat.V, flat.F, flat.E, int(flat.var_edge.size)
        for name in ('fac_ptr', 'edge_var', 'edge_fac', 'edge_canon', 'var_ptr', 'var_edge', 'edge_count', 'fac_pot',
                     'var_value', 'var_dom', 'var_mult', 'fac_mult', 'dom_cont', 'dom_lo', 'dom_hi', 'dom_ptr',
                     'dom_val'):
            setattr(g, name, ptr(t[name]))
        g.D = int(flat.dom_cont.size)
        g.edge_value, g.slot_var = ptr(t['edge_value']), ptr(t['slot_var'])
        g.hub_vars, g.n_hubs = ptr(t['hub_vars']), int(hubs.size)
        self.g = g
        p = PotsStruct()
        p.P = int(flat.pot_kind.size)
        p.kind, p.off, p.param = ptr(t['pot_kind']), ptr(t['pot_off']), ptr(t['pot_param'])
        # formulas the device has to interpret (lhvi_pots_t.interpreted): hard formulas, and soft ones without a
        # conditional-quadratic block (word 2 of the row, lhvi/mln.py::device_spec)
        p.interpreted = int(interpreted)
        self.p = p

    def zeros(self, *shape, dtype=None):
        torch = _torch()
        return torch.zeros(*shape, dtype=dtype or torch.float64, device=self.device)

    def empty(self, *shape, dtype=None):
        torch = _torch()
        return torch.empty(*shape, dtype=dtype or torch.float64, device=self.device)
