"""ctypes binding of the C-ABI shared library (include/sttode_hip.h).

The product path has NO fallback: if ``libsttode_hip.so`` is missing or a call fails, this raises.
Pointers are raw device addresses (``tensor.data_ptr()``); kernels are enqueued on the caller's
current HIP stream (``torch.cuda.current_stream().cuda_stream``).
"""
import ctypes
import os

_LIB = None
LIB_PATH = os.environ.get('STTODE_HIP_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'lib', 'libsttode_hip.so')

_P, _I, _L, _F, _D = ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_float, ctypes.c_double
ABI_VERSION = 7   # == STTODE_ABI_VERSION of include/sttode_hip.h; lib() refuses a library built from another header

# name -> argtypes (mirrors include/sttode_hip.h; tests/test_capi_symbols.py checks header == table == .so)
SIGNATURES = {
    'sttode_abi_version': [],
    'sttode_rotate_scene': [_P, _P, _I, _I, _I, _F, _F, _P],
    'sttode_frontend_scenes': [_P, _P, _I, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P],
    'sttode_frontend_nba': [_P, _I, _I, _I, _I, _P, _P, _P, _P, _P, _P],
    'sttode_frontend_future': [_P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _P],
    'sttode_embed_qkv': [_P] * 11 + [_P, _P, _P, _P, _I, _I, _P],
    'sttode_mhgsa_attn': [_P, _P, _P, _P, _P, _P, _I, _I, _I] + [_L] * 8 + [_F, _F, _P],
    'sttode_mhgsa_attn_groups': [_P, _P, _P, _P, _I, _L, _L, _L, _L, _I, _I, _I] + [_L] * 8 + [_F, _F, _I, _P],
    'sttode_post_attn': [_P] * 14 + [_P, _P, _I, _P, _I, _F, _P],
    'sttode_post_attn_ode': [_P] * 16 + [_P, _P, _I, _F, _I, _I, _P],
    'sttode_post_attn_rhs': [_P] * 14 + [_P, _P, _I, _P, _I, _P],
    'sttode_ode_state_to_pf': [_P, _P, _P, _I, _P],
    'sttode_gru_cols': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    'sttode_set_latency_tiles': [_I, _I, _I],
    'sttode_linear_cols': [_P, _I, _I, _P, _I, _I, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_agent_preact': [_P] * 11 + [_I, _P],
    'sttode_mlp_block0': [_P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_mlp_block1': [_P, _P, _I, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P],
    'sttode_mlp_cols': [_P, _P, _I, _P, _P, _P, _I, _I, _I, _P],
    'sttode_traj_chain': [_P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _I, _I, _P],
    'sttode_chain_prog_len': [_I, _I],
    'sttode_gru_cols32': [_P, _I, _P, _P, _I, _P, _P, _I, _I, _P],
    'sttode_best_of_k': [_P, _P, _I, _I, _I, _F, _P, _P, _P],
    'sttode_horizon_metrics': [_P, _P, _I, _I, _I, _F, _P, _P],
    # stage-2 sampler (csrc/sampler.hip)
    'sttode_sampler_latent': [_P, _P, _P, _I, _P, _P, _I, _I, _I, _P],
    'sttode_sampler_loss': [_P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P],
    'sttode_sampler_loss_bwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _F, _P, _P, _P, _P],
    # training step (csrc/train.hip)
    'sttode_tlinear': [_P, _L, _I, _P, _L, _I, _P, _P, _L, _P, _L, _I, _I, _I, _I, _I, _P],
    'sttode_tlinear_tab': [_P, _L, _P, _L, _P, _P, _L, _I, _P, _L, _I, _I, _I, _I, _P],
    'sttode_twgrad': [_P, _L, _P, _L, _I, _P, _L, _P, _I, _I, _I, _P, _L, _P],
    'sttode_twgrad_defer': [_I, _P, _L],
    'sttode_twgrad_flush': [],
    'sttode_tgemm_group': [_I],
    'sttode_tlinear_bwd': [_P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _P, _L, _I, _P, _L, _P, _I, _I, _I, _P, _L, _P],
    'sttode_decoder_inputs': [_P, _P, _L, _P, _L, _P, _P, _I, _I, _I, _I, _P],
    'sttode_rows_copy': [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    'sttode_rows_reduce': [_P, _L, _P, _L, _I, _I, _I, _I, _P],
    'sttode_train_ewise': [_I, _P, _P, _P, _P, _P, _L, _I, _F, _P],
    'sttode_add_ln_fwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    'sttode_ln_bwd': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _P, _L, _P],
    'sttode_gru_cell_fwd': [_P, _L, _P, _P, _P, _P, _I, _P],
    'sttode_gru_cell_bwd': [_P, _P, _P, _P, _L, _P, _P, _I, _P],
    'sttode_gru_seq_fwd': [_P, _P, _P, _P, _P, _P, _L, _I, _I, _P],
    'sttode_gru_seq_bwd': [_P, _L, _P, _P, _P, _P, _P, _I, _I, _P],
    'sttode_conv_fwd': [_P, _I, _P, _P, _P, _P, _P, _I, _I, _P],
    'sttode_conv_bwd': [_P, _P, _P, _P, _P, _P, _I, _I, _P, _L, _P],
    'sttode_mhgsa_attn_bwd': [_P, _P, _P, _I, _I, _I, _P],
    'sttode_adam_step': [_P, _I, _L, _P, _D, _D, _D, _D, _D, _L, _P],
    'sttode_loss_sqerr': [_P, _P, _L, _F, _P, _P, _P],
    'sttode_loss_kl': [_P, _P, _I, _I, _I, _F, _F, _P, _P, _P, _P],
    'sttode_loss_diverse': [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P],
    'sttode_ttrunk_fwd': [_P, _I, _I, _I, _L, _F, _I, _P],
    'sttode_loss_objective': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _L, _P],
    # manifold op library (csrc/pmath.hip)
    'sttode_pmath_rowop': [_I, _P, _P, _P, _P, _I, _I, _F, _P],
    'sttode_pmath_scalar': [_I, _P, _P, _L, _P],
    'sttode_pmath_riemannian_grad': [_P, _P, _P, _I, _I, _F, _P],
    'sttode_pmath_matvec': [_P, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    'sttode_pmath_pair': [_I, _P, _P, _P, _P, _I, _I, _I, _F, _P],
    'sttode_pmath_mean': [_P, _P, _P, _P, _I, _I, _F, _P],
    'sttode_oblique_dist': [_P, _P, _P, _I, _I, _I, _I, _P],
    # native pipeline (csrc/pipeline.hip)
    'sttode_model_create': [ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_void_p), _I, _I, _I, _I, _I, _I],
    'sttode_model_destroy': [_P],
    'sttode_model_set_weight': [_P, _I, _P],
    'sttode_workspace_layout': [_P, _I, _I, ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)],
    'sttode_workspace_init': [_P, _P, _I, _I, _P],
    'sttode_timeout_word': [_P, ctypes.POINTER(ctypes.c_void_p)],
    'sttode_timeout_clear': [_P],
    'sttode_set_col_parts': [_P, _I],
    'sttode_set_chain': [_P, _I],
    'sttode_set_fused': [_P, _I],
    'sttode_set_mfma_mode': [_P, _I],
    'sttode_debug_drop_role_flag': [_P, _I],
    'sttode_set_scene_launch': [_P, _I],
    'sttode_stage_scene': [_P, _P, _I, _I, _I, _P, _P],
    'sttode_stage_rows': [_P, _L, _P, _L, _P, _P],
    'sttode_loss_objective_live': [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _F, _F, _P, _P, _P, _P, _P, _P, _L, _P],
    'sttode_live_rows_gather': [_P, _I, _P, _I, _I, _P],
    'sttode_publish_values': [_P, _I, _P, _P, _P, _P],
    'sttode_wait_value': [_P, _L, _D],
    'sttode_async_next_stream': [_P, _I, _P],
    'sttode_async_best_of_k': [_P, _I, _P, _P, _I, _I, _I, _F, _P, _P],
    'sttode_fused_block_of': [_L, _L, _L, _L, _L],
    'sttode_set_ode': [_P, _I, _I],
    'sttode_timing_enable': [_P, _I],
    'sttode_timing_read': [_P, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_double)],
    'sttode_inference_scenes': [_P, _P, _P, _I, _I, _P, _P, _P, _P],
    'sttode_inference_nba': [_P, _P, _I, _I, _P, _P, _P, _P],
    'sttode_inference_nba_groups': [_P, _P, _I, _I, _I, _P, _P, _P, _P],
    'sttode_inference_scenes_async': [_P, _P, _P, _I, _I, _P, _P, _P, _I, _P, _P],
    'sttode_inference_nba_async': [_P, _P, _I, _I, _P, _P, _P, _I, _P, _P],
    'sttode_async_horizon_metrics': [_P, _I, _P, _P, _I, _I, _I, _F, _P],
    'sttode_wait': [_P, _I, _P],
    'sttode_async_is_lagged': [_P, _I],
    'sttode_wait_host': [_P, _I],
    'sttode_set_lagged': [_P, _I],
    'sttode_async_flush': [_P],
    'sttode_clock_probe': [_P, _P],
    'sttode_copy_to_host': [_P, _P, _L, _I, _P],
    'sttode_async_enqueue': [_P, _I],
    'sttode_check': [_P, _P, _I, _I, _P],
}

# enum SttodeWeight / SttodeBuffer / SttodeStage of include/sttode_hip.h (order is ABI)
WEIGHT_ORDER = ([('past', k) for k in ('fc1P', 'fc1b', 'posP', 'peb', 'fc2P', 'fc2b', 'fc3P', 'fc3b', 'fc3last', 'inP', 'inb', 'outP',
                                       'outb', 'infoP', 'infob', 'gateP', 'gateb', 'ln1w', 'ln1b', 'l1P', 'l1b', 'l2P', 'l2b', 'ln2w',
                                       'ln2b')]
                + [('blk0', k) for k in ('convP', 'convB', 'wihP', 'whhP', 'gbias', 'x_WA', 'x_b1', 'y_WA', 'y_b1', 'stream')]
                + [('blk1', k) for k in ('convP', 'convB', 'wihP', 'whhP', 'gbias', 'y_WA', 'y_b1', 'stream')]
                + [('chain', k) for k in ('pool', 'prog', 'consts')] + [('gru0s', k) for k in ('pool', 'prog', 'consts')]
                + [('chain_b3', k) for k in ('pool', 'prog')]
                + [('role32', k) for k in ('pool', 'prog_scenes', 'prog_nba', 'consts_scenes', 'consts_nba')])
TRUNK_PTRS = ('fc1_w', 'fc1_b', 'pos_w', 'pos_b', 'fc2_w', 'fc2_b', 'fc3_w', 'fc3_b', 'inproj_w', 'inproj_b', 'out_w', 'out_b', 'info_w', 'info_b',
              'gate_w', 'gate_b', 'ln1_w', 'ln1_b', 'l1_w', 'l1_b', 'l2_w', 'l2_b', 'ln2_w', 'ln2_b', 'enc_in', 'last', 'pe', 'drop', 'posin', 'tp',
              'h3in', 'feat', 'xc', 'qkv', 'ao', 'tt', 'ss', 'h', 'xh1', 'rs1', 'f1', 'xh2', 'rs2', 'ode', 'attn')   # enum SttodeTrunkPtr
BUFFERS = ('scene_orig', 'agent_scene', 'xpad', 'enc_in', 'cur', 'orig', 'last', 'g', 'qkv', 'attn', 'pf', 'state0', 'A0x', 'A0y',
           'A1y', 'dbuf', 'ybuf', 'state1', 'queue', 'flags', 'ode')
STAGES = ('frontend', 'embed_qkv', 'mhgsa_attn', 'post_attn', 'gru_cols[block0,agents]', 'agent_preact', 'mlp_block0',
          'gru_cols[block1,trajectories]', 'mlp_block1', 'trajectory_chain', 'agents_fused[encoder+block0 GRU]', 'agents+trajectory_chain[fused launch]')


class AsyncOpts(ctypes.Structure):
    """struct SttodeAsyncOpts of include/sttode_hip.h: everything an asynchronous call needs travels with the call."""
    _fields_ = [('device_latents', ctypes.c_int), ('zkey', ctypes.c_ulonglong), ('metrics_gt', ctypes.c_void_p), ('ade', ctypes.c_void_p),
                ('fde', ctypes.c_void_p), ('metrics_scale', ctypes.c_float), ('nba_groups', ctypes.c_int)]


class NativeModel:
    """Owner of a SttodeModel handle (csrc/pipeline.hip) plus the packed weight tensors it points into."""

    def __init__(self, packed, Tp, Tf, K):
        self.packed = packed  # keeps the device tensors alive
        tbl = (ctypes.c_void_p * len(WEIGHT_ORDER))(*[(packed[g][k].data_ptr() if g in packed else None) for g, k in WEIGHT_ORDER])
        h = ctypes.c_void_p()
        rc = lib().sttode_model_create(ctypes.byref(h), tbl, len(WEIGHT_ORDER), Tp, Tf, K, int(packed['blk0']['n_chunks']),
                                       int(packed['blk1']['n_chunks']))
        if rc != 0:
            raise SttodeError('sttode_model_create failed: ' + lib().sttode_last_error().decode())
        self.h = h
        self._layouts = {}
        w = ctypes.c_void_p()
        if lib().sttode_timeout_word(h, ctypes.byref(w)) != 0 or not w.value:
            raise SttodeError('sttode_timeout_word failed: ' + lib().sttode_last_error().decode())
        self.timeout_word = ctypes.c_uint.from_address(w.value)   # the model's host-visible time-out word: a plain host load per check

    def raise_if_timed_out(self):
        """Raise if any launch of this model gave up on its in-launch hand-off since the last check (include/sttode_hip.h: the word in
        pinned host memory is set by the group that gives up; no stream operation, no synchronisation here)."""
        v = self.timeout_word.value
        if v:
            lib().sttode_timeout_clear(self.h)
            raise SttodeError('a trajectory group of an earlier launch gave up waiting for its per-agent role (time-out word %d): the '
                              'predictions of that call are NaN-poisoned, not valid' % v if v != 2 else
                              'an earlier launch ran on a workspace that was never initialised (sttode_workspace_init): its predictions are NaN')

    def init_workspace(self, buf, n, S):
        call('sttode_workspace_init', self.h, buf, int(n), int(S), stream_ptr())

    def layout(self, n, S):
        key = (n, S)
        if key not in self._layouts:
            off = (ctypes.c_long * len(BUFFERS))()
            tot = ctypes.c_long()
            rc = lib().sttode_workspace_layout(self.h, n, S, off, ctypes.byref(tot))
            if rc != 0:
                raise SttodeError('sttode_workspace_layout failed: ' + lib().sttode_last_error().decode())
            self._layouts[key] = (dict(zip(BUFFERS, list(off))), int(tot.value))
        return self._layouts[key]

    def set_col_parts(self, parts):
        if lib().sttode_set_col_parts(self.h, int(parts)) != 0:
            raise SttodeError('sttode_set_col_parts failed: ' + lib().sttode_last_error().decode())

    def set_chain(self, mode):
        """1: fused per-trajectory chain kernel, 0: three-kernel form, -1: automatic."""
        if lib().sttode_set_chain(self.h, int(mode)) != 0:
            raise SttodeError('sttode_set_chain failed: ' + lib().sttode_last_error().decode())

    def set_scene_launch(self, max_tiles):
        """Serial scene calls with at most `max_tiles` 16-trajectory tiles run as ONE launch (csrc/scene_lat.hip); -1: default (128),
        0: never (the six-launch form).  Bitwise the same predictions either way."""
        call('sttode_set_scene_launch', self.h, int(max_tiles))

    def set_fused(self, mode):
        """1: per-agent roles inside the chain launch (default); 2: the roles also run the scene front-end (one launch per call);
        3: as 1 with the roles interleaved 160 groups ahead of their consumers in the grid; 4: as 1 with five role workgroups per tile
        (E | G | three tables) instead of one; 0: separate per-agent launches.
        Results are bitwise the same in every mode."""
        if lib().sttode_set_fused(self.h, int(mode)) != 0:
            raise SttodeError('sttode_set_fused failed: ' + lib().sttode_last_error().decode())

    def set_lagged(self, streams):
        """Pipelined calls in the LAGGED form (default 3 streams): a call's launch = its throughput-form per-agent roles + the trajectory
        groups of the call made `streams` calls earlier; 0: the round-3 forms (bitwise the serial forms).  See include/sttode_hip.h."""
        call('sttode_set_lagged', self.h, int(streams))
        self._lagged = int(streams)

    def check(self, workspace, n, S):
        """Raise if a group of the last launch on `workspace` gave up waiting for its producer (in-launch hand-off forms only)."""
        call('sttode_check', self.h, workspace, int(n), int(S), stream_ptr())

    def set_weights(self, group, tensors):
        """Hand a weight group that was not there at creation (the opt-in bf16-split stream) to the native model."""
        self.packed[group] = tensors
        for i, (g, k) in enumerate(WEIGHT_ORDER):
            if g == group:
                call('sttode_model_set_weight', self.h, i, tensors[k])

    def set_mfma_mode(self, mode):
        """EXPLORATORY: 1 = block-0 decoder MLPs of the fused launch as a three-way bf16 split on the bf16 matrix cores; 0 = fp32."""
        if lib().sttode_set_mfma_mode(self.h, int(mode)) != 0:
            raise SttodeError('sttode_set_mfma_mode failed: ' + lib().sttode_last_error().decode())

    def set_ode(self, method, steps):
        if lib().sttode_set_ode(self.h, int(method), int(steps)) != 0:
            raise SttodeError('sttode_set_ode failed: ' + lib().sttode_last_error().decode())

    def timing(self, every):
        """0/False: off; n: bracket every n-th forward call (True == every call)."""
        lib().sttode_timing_enable(self.h, int(every))

    def read_timing(self):
        ms = (ctypes.c_double * len(STAGES))()
        cnt = (ctypes.c_int * len(STAGES))()
        busy = (ctypes.c_double * len(STAGES))()
        rc = lib().sttode_timing_read(self.h, ms, cnt, busy)
        if rc != 0:
            raise SttodeError('sttode_timing_read failed: ' + lib().sttode_last_error().decode())
        self.busy_ms = {STAGES[i]: busy[i] for i in range(len(STAGES)) if cnt[i]}     # union of the launches' intervals per stage
        return {STAGES[i]: (ms[i], cnt[i]) for i in range(len(STAGES)) if cnt[i]}

    def __del__(self):
        try:
            if self.h:
                lib().sttode_model_destroy(self.h)
                self.h = None
        except Exception:
            pass



class SttodeError(RuntimeError):
    pass


def lib():
    """Load (once) and return the shared library; raise loudly if it is not built."""
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise SttodeError(f'{LIB_PATH} not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              f'or `make -C sttode_amd/csrc`. There is no CPU fallback.')
        L = ctypes.CDLL(LIB_PATH)
        L.sttode_last_error.restype = ctypes.c_char_p
        L.sttode_last_error.argtypes = []
        try:
            got = int(L.sttode_abi_version())
        except AttributeError:
            got = None
        if got != ABI_VERSION:   # a stale or variant build (STTODE_HIP_LIB): its enums / signatures would be indexed with this file's tables
            raise SttodeError(f'{LIB_PATH}: ABI version {got}, this binding needs {ABI_VERSION}; rebuild with `make -C sttode_amd/csrc`')
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = ctypes.c_int
        _LIB = L
    return _LIB


def _ptr(t):
    if t is None:
        return None
    return t.data_ptr() if hasattr(t, 'data_ptr') else int(t)


# When set to a list, every call is bracketed by HIP events recorded on the launch stream (torch's current
# stream == the stream handed to the kernels); bench.py uses this for per-kernel durations in the timed region.
TIMING = None


def call(name, *args, tag=None):
    """Invoke an entry point; tensors are converted to device pointers; non-zero status raises."""
    L = lib()
    conv = [(_ptr(a) if (a is None or hasattr(a, 'data_ptr')) else a) for a in args]
    if TIMING is not None:
        import torch
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = getattr(L, name)(*conv)
        e1.record()
        TIMING.append((tag or name, e0, e1))
    else:
        rc = getattr(L, name)(*conv)
    if rc != 0:
        raise SttodeError(f'{name} failed (status {rc}): {L.sttode_last_error().decode()}')


def stream_ptr():
    """Raw hipStream_t of torch's current stream on the current device (the C accessor: torch.cuda.current_stream() builds a Stream object
    through three Python layers, ~9 us, twice per one-scene call)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:
        return torch.cuda.current_stream().cuda_stream
