"""Host-side weight packing into MFMA fragment order (done once per weight set).

PK16 (see csrc/chain.hpp): for W [N, K] (N, K padded to multiples of 16)
    P[it, T, lane, r] = W[16*it + (lane & 15), 16*T + 4*(lane >> 4) + r]
so that a wave's A fragment for four consecutive v_mfma_f32_16x16x4_f32 is one contiguous 1 KiB read.

Everything here is a pure re-layout of the reference's parameters (state_dict names in SURVEY.md §8b)
plus three algebraic foldings that do not change the function computed:
  * pos-encoder:  fc(cat(x, pe[t])) = W[:, :64] x + (W[:, 64:] pe[t] + b)          (model/STTODE.py:170-173)
  * category:     input_fc3(cat(f, onehot)) = W[:, :64] f + b + [last agent] W[:, 66] (model/STTODE.py:199-210,223)
  * decoder L1:   W1 [pf | z | state] = W1[:, pf] pf + W1[:, z] z + W1[:, state] state  (model/STTODE.py:71,74-75)
"""
import numpy as np


def _pad16(n):
    return (n + 15) // 16 * 16


def pk16(W):
    """[N, K] -> float32 [N/16, K/16, 64, 4] (zero padded)."""
    W = np.asarray(W, np.float32)
    N, K = W.shape
    Np, Kp = _pad16(N), _pad16(K)
    Wp = np.zeros((Np, Kp), np.float32)
    Wp[:N, :K] = W
    X = Wp.reshape(Np // 16, 16, Kp // 16, 4, 4)          # it, i, T, q, r
    X = X.transpose(0, 2, 3, 1, 4)                         # it, T, q, i, r   (lane = q*16 + i)
    return np.ascontiguousarray(X.reshape(Np // 16, Kp // 16, 64, 4))


def pk4(W):
    """[N, 4] (K = 4: a single MFMA) -> [N/16, 64] with lane(i, q) = W[16*it + i, q]."""
    W = np.asarray(W, np.float32)
    N = W.shape[0]
    assert W.shape[1] == 4 and N % 16 == 0
    return np.ascontiguousarray(W.reshape(N // 16, 16, 4).transpose(0, 2, 1).reshape(N // 16, 64))


def pad_vec(b, n):
    out = np.zeros(n, np.float32)
    out[: len(b)] = b
    return out


def toeplitz_conv(w, Tp, TPX):
    """Conv1d(2->32, k=3, pad=1) over Tp frames as a [Tp*32, 16*TPX] matrix acting on the flattened
    (t, c) sequence: out[t*32 + o] = sum_{c,k} w[o, c, k] * x[c, t + k - 1]   (model/STTODE.py:30,65)."""
    w = np.asarray(w, np.float32)
    M = np.zeros((Tp * 32, 16 * TPX), np.float32)
    for t in range(Tp):
        for k in range(3):
            tp = t + k - 1
            if 0 <= tp < Tp:
                for c in range(2):
                    M[t * 32:(t + 1) * 32, 2 * tp + c] = w[:, c, k]
    return M


def mlp_stream(W1v, W2, W3p, b2, b3p):
    """Weight stream of ONE MLP for csrc/decoder.hip mlp_phase (one 16-row hidden tile per chunk):
    32 "L12" chunks { W1v tiles [KTV] , W2 tiles [16] } followed by N3 "L3" chunks { W3 tiles [TP3][16] | b3 of those tiles |
    (first L3 chunk only) b2 } -- the biases ride in the spare space of the L3 chunks and reach LDS by the same DMA.
    Returns a float32 array [n_chunks, CHW*4]."""
    P1 = pk16(W1v)             # [32, KTV, 64, 4]
    P2 = pk16(W2)              # [16, 32, 64, 4]
    P3 = pk16(W3p)             # [NO, 16, 64, 4]
    assert P1.shape[0] == 32 and P2.shape[:2] == (16, 32) and P3.shape[1] == 16
    KTV, NO = P1.shape[1], P3.shape[0]
    chw = (KTV + 16) * 64 * 4                # floats per chunk
    tp3 = (chw // 4) // (16 * 64)            # layer-3 output tiles per chunk
    assert tp3 * 16 * 64 * 4 + tp3 * 16 + 256 <= chw, 'no room for the biases in the L3 chunk'
    assert len(b2) == 256 and len(b3p) == 16 * NO
    chunks = []
    for ch in range(32):
        chunks.append(np.concatenate([P1[ch].reshape(-1), P2[:, ch].reshape(-1)]))
    for c3 in range((NO + tp3 - 1) // tp3):
        buf = np.zeros(chw, np.float32)
        part = P3[c3 * tp3:(c3 + 1) * tp3].reshape(-1)
        buf[: part.size] = part
        o = tp3 * 16 * 64 * 4
        bb = np.asarray(b3p[16 * c3 * tp3: 16 * (c3 + 1) * tp3], np.float32)
        buf[o: o + bb.size] = bb
        if c3 == 0:
            buf[o + 16 * tp3: o + 16 * tp3 + 256] = b2
        chunks.append(buf)
    out = np.stack(chunks)
    assert out.shape[1] == chw
    return np.ascontiguousarray(out)


def tiles_x(Tp):
    return 1 if 2 * Tp <= 16 else 2


def tiles_y(Tf):
    return (2 * Tf + 15) // 16


SUPPORTED_NOY = (1, 2, 3, 5)


def pack_trunk(sd, prefix, Tlen):
    """Encoder trunk (PastEncoder / FutureEncoder shared part)."""
    g = lambda k: np.asarray(sd[prefix + k], np.float32)
    att = 'ODE_Encoder.odeblock.odefunc.layers.0.'
    Wpos = g('pos_encoder.fc.weight')
    pe = g('pos_encoder.pe')
    W3 = g('input_fc3.weight')
    return {
        'fc1P': pk4(g('input_fc.weight')), 'fc1b': g('input_fc.bias'),
        'posP': pk16(Wpos[:, :64]),
        'peb': np.ascontiguousarray(pe[:Tlen] @ Wpos[:, 64:].T + g('pos_encoder.fc.bias')),
        'fc2P': pk16(g('input_fc2.weight')), 'fc2b': g('input_fc2.bias'),
        'fc3P': pk16(W3[:, :64]), 'fc3b': g('input_fc3.bias'), 'fc3last': np.ascontiguousarray(W3[:, 66]),
        'inP': pk16(g(att + 'self_attn.temporal_attention_before.in_proj_weight')),
        'inb': g(att + 'self_attn.temporal_attention_before.in_proj_bias'),
        'outP': pk16(g(att + 'self_attn.temporal_attention_before.out_proj.weight')),
        'outb': g(att + 'self_attn.temporal_attention_before.out_proj.bias'),
        'infoP': pk16(g(att + 'self_attn.temporal_info.weight')), 'infob': g(att + 'self_attn.temporal_info.bias'),
        'gateP': pk16(g(att + 'self_attn.temporal_gate.weight')), 'gateb': g(att + 'self_attn.temporal_gate.bias'),
        'ln1w': g(att + 'norm1.weight'), 'ln1b': g(att + 'norm1.bias'),
        'l1P': pk16(g(att + 'linear1.weight')), 'l1b': g(att + 'linear1.bias'),
        'l2P': pk16(g(att + 'linear2.weight')), 'l2b': g(att + 'linear2.bias'),
        'ln2w': g(att + 'norm2.weight'), 'ln2b': g(att + 'norm2.bias'),
    }


def pack_block(sd, i, Tp, Tf, first):
    """DecomposeBlock i.  ``first``: block 0 (state term is per-agent, z term per-trajectory, x and y MLPs);
    otherwise block >= 1 (state per trajectory; the last block's decoder_x is dead in inference)."""
    p = f'decoder.decompose.{i}.'
    g = lambda k: np.asarray(sd[p + k], np.float32)
    TPX, NOY = tiles_x(Tp), tiles_y(Tf)
    bih, bhh = g('encoder_past.bias_ih_l0'), g('encoder_past.bias_hh_l0')
    # Gate rows are pre-scaled so the kernel's sigmoid / tanh need no multiply (csrc/chain.hpp *_prescaled):
    #   r, z rows (0..191) by -log2(e):   sigmoid(x) = 1 / (1 + 2^(-x log2 e))
    #   n rows (192..287)  by 2 log2(e):  tanh(x)    = 1 - 2 / (1 + 2^(2 x log2 e));  x = a_in + r * a_hn is linear in both parts
    L2E = np.float32(1.4426950408889634)
    sc = np.concatenate([np.full(192, -L2E, np.float32), np.full(96, 2 * L2E, np.float32)])
    wih = g('encoder_past.weight_ih_l0') * sc[:, None]
    whh = g('encoder_past.weight_hh_l0') * sc[:, None]
    out = {
        'convP': pk16(toeplitz_conv(g('conv_past.weight'), Tp, TPX)), 'convB': g('conv_past.bias'),
        'wihP': pk16(wih), 'whhP': pk16(whh),
        'gbias': np.ascontiguousarray(np.stack([(bih[:96] + bhh[:96]) * -L2E, (bih[96:192] + bhh[96:192]) * -L2E,
                                                bih[192:] * (2 * L2E), bhh[192:] * (2 * L2E)]).astype(np.float32)),
    }
    streams = []
    for nm, NO in ((('x', TPX), ('y', NOY)) if first else (('y', NOY),)):
        W1, b1 = g(f'decoder_{nm}.layers.0.weight'), g(f'decoder_{nm}.layers.0.bias')
        W2, b2 = g(f'decoder_{nm}.layers.1.weight'), g(f'decoder_{nm}.layers.1.bias')
        W3, b3 = g(f'decoder_{nm}.layers.2.weight'), g(f'decoder_{nm}.layers.2.bias')
        W3p = np.zeros((16 * NO, 256), np.float32)
        W3p[: W3.shape[0]] = W3
        if first:
            out[nm + '_WA'] = pk16(np.concatenate([W1[:, :128], W1[:, 160:]], axis=1))   # [pf | state0] per agent
            streams.append(mlp_stream(W1[:, 128:160], W2, W3p, b2, pad_vec(b3, 16 * NO)))  # z per trajectory
        else:
            out[nm + '_WA'] = pk16(W1[:, :128])                                           # pf per agent
            streams.append(mlp_stream(W1[:, 128:], W2, W3p, b2, pad_vec(b3, 16 * NO)))    # [z | state] per trajectory
        out[nm + '_b1'] = b1
    out['stream'] = np.ascontiguousarray(np.concatenate(streams, axis=0))
    out['n_chunks'] = out['stream'].shape[0]
    if not first:
        # decoder_x of a non-first block: dead in inference(), needed by forward() (recover_traj, model/STTODE.py:339-341)
        W1, W2, W3 = g('decoder_x.layers.0.weight'), g('decoder_x.layers.1.weight'), g('decoder_x.layers.2.weight')
        W3p = np.zeros((16 * TPX, 256), np.float32)
        W3p[: W3.shape[0]] = W3
        out['x_WA'] = pk16(W1[:, :128])
        out['x_b1'] = g('decoder_x.layers.0.bias')
        out['x_stream'] = mlp_stream(W1[:, 128:], W2, W3p, g('decoder_x.layers.1.bias'), pad_vec(g('decoder_x.layers.2.bias'), 16 * TPX))
        out['x_n_chunks'] = out['x_stream'].shape[0]
    return out


# ---------------------------------------------------------------------------------------------------
# PK32 / fused trajectory chain (csrc/chain32.hip)
# ---------------------------------------------------------------------------------------------------
def pk32_tile(Wb):
    """[32, 32] block (rows = output features, columns = k) -> 1024 floats in v_mfma_f32_32x32x2_f32 A-operand order:
    T[g, lane, r] = Wb[lane & 31, 8*g + 4*(lane >> 5) + r]  (four ds_read_b128 per lane feed 16 MFMAs; MFMA step 4g+r
    consumes the k-pair (8g + r, 8g + 4 + r), which is where the 32x32 accumulator layout keeps those features)."""
    Wb = np.asarray(Wb, np.float32)
    assert Wb.shape == (32, 32)
    X = Wb.reshape(32, 4, 2, 4).transpose(1, 2, 0, 3)     # i, g, h, r -> g, h, i, r   (lane = 32*h + i)
    return np.ascontiguousarray(X).reshape(-1)


def pk32_tiles(W):
    """[N, K] (zero padded to multiples of 32) -> [N/32, K/32, 1024]."""
    W = np.asarray(W, np.float32)
    N, K = W.shape
    Np, Kp = (N + 31) // 32 * 32, (K + 31) // 32 * 32
    Wp = np.zeros((Np, Kp), np.float32)
    Wp[:N, :K] = W
    return np.stack([np.stack([pk32_tile(Wp[32 * i:32 * i + 32, 32 * j:32 * j + 32]) for j in range(Kp // 32)]) for i in range(Np // 32)])


def _bf16_rne(x):
    """float32 array -> (bf16 bit patterns uint16, the same values widened back to float32); round to nearest even."""
    u = np.ascontiguousarray(x, np.float32).view(np.uint32)
    r = ((u + np.uint32(0x7fff) + ((u >> np.uint32(16)) & np.uint32(1))) >> np.uint32(16)).astype(np.uint16)
    return r, (r.astype(np.uint32) << np.uint32(16)).view(np.float32)


def pk32b_tile(Wb):
    """[32, 32] block -> 1536 float32-sized words holding the THREE-WAY bf16 SPLIT of the block in v_mfma_f32_32x32x16_bf16 A-operand
    order (exploratory mode, csrc/chain32.hip B3): W = hi + mid + lo with 8 mantissa bits each (round to nearest even at every level).
    Layout [k block kb (2)][plane (3: hi, mid, lo)][lane (64)][8 bf16]: lane l holds row l & 31, k slot s of half h = l >> 5 is
    column 16 kb + 8 (s // 4) + 4 h + s % 4 -- the feature that accumulator register 8 kb + s of the producing layer holds for that lane,
    so the fp32 accumulator layout of a layer is still the B layout of the next."""
    Wb = np.asarray(Wb, np.float32)
    assert Wb.shape == (32, 32)
    hi_b, hi_f = _bf16_rne(Wb)
    mi_b, mi_f = _bf16_rne(Wb - hi_f)
    lo_b, _ = _bf16_rne((Wb - hi_f) - mi_f)
    out = np.zeros((2, 3, 64, 8), np.uint16)
    lane = np.arange(64)
    row, h = lane & 31, lane >> 5
    for kb in range(2):
        for s_ in range(8):
            col = 16 * kb + 8 * (s_ // 4) + 4 * h + s_ % 4
            for pl, src in enumerate((hi_b, mi_b, lo_b)):
                out[kb, pl, :, s_] = src[row, col]
    return np.ascontiguousarray(out).reshape(-1).view(np.float32)


def pk32b_tiles(W):
    """[N, K] (zero padded to multiples of 32) -> [N/32, K/32, 1536] (pk32b_tile of every block)."""
    W = np.asarray(W, np.float32)
    N, K = W.shape
    Np, Kp = (N + 31) // 32 * 32, (K + 31) // 32 * 32
    Wp = np.zeros((Np, Kp), np.float32)
    Wp[:N, :K] = W
    return np.stack([np.stack([pk32b_tile(Wp[32 * i:32 * i + 32, 32 * j:32 * j + 32]) for j in range(Kp // 32)]) for i in range(Np // 32)])


def tiles_y32(Tf):
    return (2 * Tf + 31) // 32


def chain_prog_len(Tp, Tf):
    l3y = (8 * tiles_y32(Tf) + 2) // 3
    return (48 + 3) + (48 + l3y) + 13 * Tp + (64 + l3y)


def chain_stream_b3(sd, Tp, Tf):
    """Exploratory mode (csrc/chain32.hip, traj_chain_kernel<NY, FUSE, true>): the same consumption order as chain_stream, with every
    tile -- the three decoder MLPs and block 1's conv + GRU -- as a three-way bf16 split (pk32b_tile, 6 KiB) for the bf16 matrix cores.  One flat pool with tiles of both sizes; the program holds
    (offset in 16-byte units, number of 1-KiB pieces) per chunk of <= 3 tiles.  Layer 3 of the split MLPs is laid out k-tile major
    (every activation tile is split once and feeds all output tiles)."""
    f = chain_stream(sd, Tp, Tf)
    NY = tiles_y32(Tf)
    g = lambda k: np.asarray(sd[k], np.float32)
    words, prog = [], []
    pos = 0

    def add(ts):
        nonlocal pos
        for o in range(0, len(ts), 3):
            grp = ts[o:o + 3]
            prog.append((pos // 4, sum(t.size for t in grp) // 256))
            for t in grp:
                words.append(t)
                pos += t.size

    def mlp_b3(prefix, kcols, NO, n_out):
        W1, W2, W3 = g(prefix + 'layers.0.weight'), g(prefix + 'layers.1.weight'), g(prefix + 'layers.2.weight')
        P1, P2 = pk32b_tiles(W1[:, kcols]), pk32b_tiles(W2)
        W3p = np.zeros((32 * NO, 256), np.float32)
        W3p[:n_out] = W3
        P3 = pk32b_tiles(W3p)
        for ht in range(16):
            add(list(P1[ht]) + [P2[R, ht] for R in range(8)])
        add([P3[o, T] for T in range(8) for o in range(NO)])          # k-tile major

    mlp_b3('decoder.decompose.0.decoder_x.', slice(128, 160), 1, 2 * Tp)
    mlp_b3('decoder.decompose.0.decoder_y.', slice(128, 160), NY, 2 * Tf)
    n_b3 = len(prog)
    l3y = (8 * NY + 2) // 3
    assert n_b3 == (48 + 3) + (48 + l3y)
    # block-1 conv + GRU: gate rows pre-scaled exactly as in chain_stream, same tile order; the tiles are re-streamed every step
    p_ = 'decoder.decompose.1.'
    L2E = np.float32(1.4426950408889634)
    sc = np.concatenate([np.full(192, -L2E, np.float32), np.full(96, 2 * L2E, np.float32)])
    Pih = pk32b_tiles(g(p_ + 'encoder_past.weight_ih_l0') * sc[:, None])
    Phh = pk32b_tiles(g(p_ + 'encoder_past.weight_hh_l0') * sc[:, None])
    conv = pk32b_tiles(toeplitz_conv(g(p_ + 'conv_past.weight'), Tp, 2))
    gru = []
    for j in range(3):
        gru += [Pih[j, 0], Phh[j, 0], Phh[j, 1], Phh[j, 2], Pih[3 + j, 0], Phh[3 + j, 0], Phh[3 + j, 1], Phh[3 + j, 2],
                Phh[6 + j, 0], Phh[6 + j, 1], Phh[6 + j, 2], Pih[6 + j, 0]]
    gru_off = pos // 4
    for t_ in gru:
        words.append(t_)
        pos += t_.size
    conv_off = pos // 4
    for t in range(Tp):
        words.append(conv[t, 0])
        pos += conv[t, 0].size
    for t in range(Tp):
        prog.append((conv_off + 384 * t, 6))
        prog.extend((gru_off + 3 * 384 * c, 18) for c in range(12))
    mlp_b3('decoder.decompose.1.decoder_y.', slice(128, 256), NY, 2 * Tf)   # block-1 decoder_y: k = [z | state1]
    assert len(prog) == f['prog_len']
    return {'pool': np.ascontiguousarray(np.concatenate(words)), 'prog': np.ascontiguousarray(np.asarray(prog, np.int32)),
            'consts': f['consts'], 'prog_len': len(prog)}


def chain_stream(sd, Tp, Tf):
    """Weight stream of the fused per-trajectory chain (csrc/chain32.hip): block-0 decoder_x, block-0 decoder_y, block-1
    conv + GRU, block-1 decoder_y.  Returns
      pool   float32 [n_tiles, 1024]  PK32 tiles
      prog   int32   [n_chunks, 2]    (first tile, tile count <= 3) in the order the kernel consumes them for ONE group
      consts float32 [1216 + 64*NY]   b2x b3x | b2y b3y | GRU gate biases (pre-scaled) conv bias | b2 b3 of block-1 decoder_y
    The kernel's consumption order is fixed (chain32.hip); this function is its single source of truth on the host."""
    NY = tiles_y32(Tf)
    g = lambda k: np.asarray(sd[k], np.float32)
    tiles, prog = [], []

    def add(ts):                     # append tiles, cut into chunks of <= 3 consecutive tiles
        base = len(tiles)
        tiles.extend(ts)
        for o in range(0, len(ts), 3):
            prog.append((base + o, min(3, len(ts) - o)))

    def mlp(prefix, kcols, NO, n_out):
        W1, W2, W3 = g(prefix + 'layers.0.weight'), g(prefix + 'layers.1.weight'), g(prefix + 'layers.2.weight')
        P1 = pk32_tiles(W1[:, kcols])                     # [16, KT1, 1024]
        P2 = pk32_tiles(W2)                               # [8, 16, 1024]
        W3p = np.zeros((32 * NO, 256), np.float32)
        W3p[:n_out] = W3
        P3 = pk32_tiles(W3p)                              # [NO, 8, 1024]
        for ht in range(16):                              # per hidden tile: KT1 layer-1 tiles, then the 8 layer-2 row tiles
            add(list(P1[ht]) + [P2[R, ht] for R in range(8)])
        add([P3[o, T] for o in range(NO) for T in range(8)])
        b3 = np.zeros(32 * NO, np.float32)
        b3[:n_out] = g(prefix + 'layers.2.bias')
        return g(prefix + 'layers.1.bias'), b3

    b2x, b3x = mlp('decoder.decompose.0.decoder_x.', slice(128, 160), 1, 2 * Tp)
    b2y, b3y = mlp('decoder.decompose.0.decoder_y.', slice(128, 160), NY, 2 * Tf)
    # block-1 conv + GRU: gate rows pre-scaled as in pack_block (sigmoid / tanh without a multiply, csrc/chain.hpp)
    p = 'decoder.decompose.1.'
    L2E = np.float32(1.4426950408889634)
    sc = np.concatenate([np.full(192, -L2E, np.float32), np.full(96, 2 * L2E, np.float32)])
    wih = g(p + 'encoder_past.weight_ih_l0') * sc[:, None]
    whh = g(p + 'encoder_past.weight_hh_l0') * sc[:, None]
    bih, bhh = g(p + 'encoder_past.bias_ih_l0'), g(p + 'encoder_past.bias_hh_l0')
    Pih, Phh = pk32_tiles(wih), pk32_tiles(whh)           # [9, 1, 1024], [9, 3, 1024]; row tiles r0..2 z0..2 n0..2
    gru = []
    for j in range(3):                                    # r:[e h0 h1 h2] z:[e h0 h1 h2] n_h:[h0 h1 h2] n_i:[e]
        gru += [Pih[j, 0], Phh[j, 0], Phh[j, 1], Phh[j, 2], Pih[3 + j, 0], Phh[3 + j, 0], Phh[3 + j, 1], Phh[3 + j, 2],
                Phh[6 + j, 0], Phh[6 + j, 1], Phh[6 + j, 2], Pih[6 + j, 0]]
    conv = pk32_tiles(toeplitz_conv(g(p + 'conv_past.weight'), Tp, 2))   # [Tp, 1, 1024]: step t = rows 32t..32t+31
    gru_base, conv_base = len(tiles), len(tiles) + 36
    tiles.extend(gru)
    tiles.extend(conv[t, 0] for t in range(Tp))
    for t in range(Tp):
        prog.append((conv_base + t, 1))
        prog.extend((gru_base + 3 * c, 3) for c in range(12))
    gbias = np.concatenate([(bih[:96] + bhh[:96]) * -L2E, (bih[96:192] + bhh[96:192]) * -L2E, bih[192:] * (2 * L2E), bhh[192:] * (2 * L2E)])
    b2m, b3m = mlp(p + 'decoder_y.', slice(128, 256), NY, 2 * Tf)
    consts = np.concatenate([b2x, b3x, b2y, b3y, gbias, g(p + 'conv_past.bias'), b2m, b3m]).astype(np.float32)
    assert consts.size == 1216 + 64 * NY and len(prog) == chain_prog_len(Tp, Tf), (consts.size, len(prog))
    return {'pool': np.ascontiguousarray(np.stack(tiles)), 'prog': np.ascontiguousarray(np.asarray(prog, np.int32)),
            'consts': np.ascontiguousarray(consts), 'prog_len': len(prog)}


def gru32_stream(sd, block, Tp):
    """Weight stream of the stand-alone streaming GRU (csrc/chain32.hip gru32_kernel) for DecomposeBlock ``block``: the same tile order
    as the GRU phase of chain_stream.  Returns pool [36 + Tp, 1024], prog [13*Tp, 2], consts [416] (gate biases [4][96], conv bias)."""
    p = f'decoder.decompose.{block}.'
    g = lambda k: np.asarray(sd[p + k], np.float32)
    L2E = np.float32(1.4426950408889634)
    sc = np.concatenate([np.full(192, -L2E, np.float32), np.full(96, 2 * L2E, np.float32)])
    wih, whh = g('encoder_past.weight_ih_l0') * sc[:, None], g('encoder_past.weight_hh_l0') * sc[:, None]
    bih, bhh = g('encoder_past.bias_ih_l0'), g('encoder_past.bias_hh_l0')
    Pih, Phh = pk32_tiles(wih), pk32_tiles(whh)
    tiles = []
    for j in range(3):                                    # r:[e h0 h1 h2] z:[e h0 h1 h2] n_h:[h0 h1 h2] n_i:[e]
        tiles += [Pih[j, 0], Phh[j, 0], Phh[j, 1], Phh[j, 2], Pih[3 + j, 0], Phh[3 + j, 0], Phh[3 + j, 1], Phh[3 + j, 2],
                  Phh[6 + j, 0], Phh[6 + j, 1], Phh[6 + j, 2], Pih[6 + j, 0]]
    conv = pk32_tiles(toeplitz_conv(g('conv_past.weight'), Tp, 2))
    tiles.extend(conv[t, 0] for t in range(Tp))
    prog = []
    for t in range(Tp):
        prog.append((36 + t, 1))
        prog.extend((3 * c, 3) for c in range(12))
    gbias = np.concatenate([(bih[:96] + bhh[:96]) * -L2E, (bih[96:192] + bhh[96:192]) * -L2E, bih[192:] * (2 * L2E), bhh[192:] * (2 * L2E)])
    consts = np.concatenate([gbias, g('conv_past.bias')]).astype(np.float32)
    assert consts.size == 416 and len(prog) == 13 * Tp
    return {'pool': np.ascontiguousarray(np.stack(tiles)), 'prog': np.ascontiguousarray(np.asarray(prog, np.int32)),
            'consts': np.ascontiguousarray(consts), 'prog_len': len(prog)}


# constants of the throughput-form per-agent roles (csrc/role32.hpp R32C): float offsets into role_stream()['consts_*']
R32_CONSTS = dict(bc=0, wlast=64, bi=128, bg=192, ln1w=256, ln1b=320, l1b=384, l2b=1408, ln2w=1472, ln2b=1536, gbias=1600, convb=1984,
                  b1x=2016, b1y=2528, b11=3040, total=3552)


def role_fold(sd, Tp):
    """The affine foldings of role_stream, float64: dict with Wc [64, 32 kte], bc, wlast [64] and, per variant v in (scenes, nba),
    Ki_v / Kg_v [64, 64], bi_v / bg_v [64]."""
    f64 = lambda k: np.asarray(sd[k], np.float64)
    pe_ = 'past_encoder.'
    att = pe_ + 'ODE_Encoder.odeblock.odefunc.layers.0.self_attn.'
    W1, b1 = f64(pe_ + 'input_fc.weight'), f64(pe_ + 'input_fc.bias')
    Wpos, bpos, pe = f64(pe_ + 'pos_encoder.fc.weight'), f64(pe_ + 'pos_encoder.fc.bias'), f64(pe_ + 'pos_encoder.pe')
    W2, b2 = f64(pe_ + 'input_fc2.weight'), f64(pe_ + 'input_fc2.bias')
    W3, b3 = f64(pe_ + 'input_fc3.weight'), f64(pe_ + 'input_fc3.bias')
    W3a = W3[:, :64]
    kte = (4 * Tp + 31) // 32
    Wc = np.zeros((64, 32 * kte))
    acc = b2.copy()
    for t in range(Tp):
        W2t = W2[:, 64 * t:64 * t + 64]
        Wc[:, 4 * t:4 * t + 4] = W3a @ W2t @ Wpos[:, :64] @ W1
        acc = acc + W2t @ (Wpos[:, :64] @ b1 + Wpos[:, 64:] @ pe[t] + bpos)
    Win, bin_ = f64(att + 'temporal_attention_before.in_proj_weight'), f64(att + 'temporal_attention_before.in_proj_bias')
    Wo, bo = f64(att + 'temporal_attention_before.out_proj.weight'), f64(att + 'temporal_attention_before.out_proj.bias')
    Wi, bi = f64(att + 'temporal_info.weight'), f64(att + 'temporal_info.bias')
    Wg, bg = f64(att + 'temporal_gate.weight'), f64(att + 'temporal_gate.bias')
    WoWv, boc = Wo @ Win[128:192], Wo @ bin_[128:192] + bo
    return {'kte': kte, 'Wc': Wc, 'bc': W3a @ acc + b3, 'wlast': W3[:, 66],
            'Ki_scenes': Wi @ WoWv, 'Kg_scenes': Wg @ WoWv, 'bi_scenes': Wi @ boc + bi, 'bg_scenes': Wg @ boc + bg,
            'Ki_nba': Wi @ Wo, 'Kg_nba': Wg @ Wo, 'bi_nba': Wi @ bo + bi, 'bg_nba': Wg @ bo + bg}


def role_stream(sd, Tp):
    """Weight stream of the THROUGHPUT form of the per-agent stage (csrc/role32.hpp, round 4): 128 agents per workgroup on the 32 MFMA columns
    of each wave, every weight streamed as PK32 tiles exactly like the trajectory chain's.  Consumption order of one workgroup:
      block-0 conv + GRU (the tiles and per-step program of gru32_stream, Tp steps)  ->  E  ->  FFN  ->  the three layer-1 tables.
    Two foldings on top of pack_trunk's (role_fold: float64 on the host, rounded once; the function is unchanged -- everything between the
    encoder's inputs and the gate nonlinearity is affine in eval mode: the positional dropout is the identity, model/STTODE.py:176,214-236):
      * E, scenes (attention length 1):  g = Wc x + bc (+ [last agent] wlast),  x = the agent's Tp x 4 encoder inputs,
            Wc[:, 4t:4t+4] = W3a W2_t Wpos_x W1,   bc = W3a (sum_t W2_t (Wpos_x b1 + Wpos_pe pe_t + bpos) + b2) + b3,   wlast = W3[:, 66];
        softmax over ONE key is 1, so the attention output is out_proj(v(g)):  info = (Wi Wo Wv) g + (Wi (Wo bv + bo) + bi), gate likewise.
      * E, attention groups > 1 (NBA): g and the attention output arrive from the launches in front;  info = (Wi Wo) attn + (Wi bo + bi).
    E tiles: [g: row tile j x k-tile (scenes only)] then per row tile j: info k0 k1, gate k0 k1.  FFN: per 32-row hidden tile: linear1 k0 k1,
    linear2 row tiles 0 1.  Tables: decoder_x, decoder_y of block 0 (k = [pf | state0]: 7 k-tiles), decoder_y of block 1 (k = pf: 4), per
    32-row tile of the 512 pre-activations its k-tiles.
    Returns pool [tiles, 1024]; prog_scenes / prog_nba int32 [chunks, 2] ((first tile, tiles <= 3): one workgroup's order, chunks are runs of
    consecutive pool tiles); consts_scenes / consts_nba float32 [R32_CONSTS['total']]; kte = k-tiles of x (ceil(4 Tp / 32))."""
    f64 = lambda k: np.asarray(sd[k], np.float64)
    att = 'past_encoder.ODE_Encoder.odeblock.odefunc.layers.0.'
    F = role_fold(sd, Tp)
    kte = F['kte']
    gs = gru32_stream(sd, 0, Tp)                                # block-0 conv + GRU: its tiles lead the pool, its program leads both programs
    tiles = list(gs['pool'])
    gru_prog = [tuple(int(v) for v in e) for e in gs['prog']]

    def chunks(first, count):                                   # a run of consecutive pool tiles cut into chunks of <= 3
        return [(first + o, min(3, count - o)) for o in range(0, count, 3)]

    def e_tiles(v):
        Pi, Pg = pk32_tiles(F['Ki_' + v]), pk32_tiles(F['Kg_' + v])   # [2, 2, 1024]
        ts = []
        if v == 'scenes':
            P = pk32_tiles(F['Wc'])                              # [2, kte, 1024]
            ts += [P[j, kt] for j in range(2) for kt in range(kte)]
        for j in range(2):
            ts += [Pi[j, 0], Pi[j, 1], Pg[j, 0], Pg[j, 1]]
        return ts

    P1, P2 = pk32_tiles(f64(att + 'linear1.weight')), pk32_tiles(f64(att + 'linear2.weight'))   # [32, 2, 1024], [2, 32, 1024]
    rest = []
    for ht in range(32):
        rest += [P1[ht, 0], P1[ht, 1], P2[0, ht], P2[1, ht]]
    for blk, nm, with_state in ((0, 'x', True), (0, 'y', True), (1, 'y', False)):
        Wl = f64(f'decoder.decompose.{blk}.decoder_{nm}.layers.0.weight')
        P = pk32_tiles(np.concatenate([Wl[:, :128], Wl[:, 160:256]], axis=1) if with_state else Wl[:, :128])   # [16, 7 | 4, 1024]
        rest += [P[rt, kt] for rt in range(16) for kt in range(P.shape[1])]
    es, en = e_tiles('scenes'), e_tiles('nba')
    base_s = len(tiles)                                         # pool: [GRU | E_scenes | FFN + tables | E_nba]
    tiles += es + rest
    base_n = len(tiles)
    tiles += en
    progs = {'scenes': gru_prog + chunks(base_s, len(es) + len(rest)),
             'nba': gru_prog + chunks(base_n, len(en)) + chunks(base_s + len(es), len(rest))}
    out = {'kte': kte, 'pool': np.ascontiguousarray(np.stack([np.asarray(t, np.float32) for t in tiles]))}
    C = R32_CONSTS
    for v in ('scenes', 'nba'):
        c = np.zeros(C['total'], np.float64)
        c[C['bc']:C['bc'] + 64], c[C['wlast']:C['wlast'] + 64] = F['bc'], F['wlast']
        c[C['bi']:C['bi'] + 64], c[C['bg']:C['bg'] + 64] = F['bi_' + v], F['bg_' + v]
        c[C['ln1w']:C['ln1w'] + 64], c[C['ln1b']:C['ln1b'] + 64] = f64(att + 'norm1.weight'), f64(att + 'norm1.bias')
        c[C['l1b']:C['l1b'] + 1024], c[C['l2b']:C['l2b'] + 64] = f64(att + 'linear1.bias'), f64(att + 'linear2.bias')
        c[C['ln2w']:C['ln2w'] + 64], c[C['ln2b']:C['ln2b'] + 64] = f64(att + 'norm2.weight'), f64(att + 'norm2.bias')
        c[C['gbias']:C['gbias'] + 416] = gs['consts']
        for nm, blk, key in (('x', 0, 'b1x'), ('y', 0, 'b1y'), ('y', 1, 'b11')):
            c[C[key]:C[key] + 512] = f64(f'decoder.decompose.{blk}.decoder_{nm}.layers.0.bias')
        out['consts_' + v] = np.ascontiguousarray(c.astype(np.float32))
        out['prog_' + v] = np.ascontiguousarray(np.asarray(progs[v], np.int32))
        out['prog_len_' + v] = len(progs[v])
        assert len(progs[v]) == role_prog_len(Tp, v == 'nba')
    return out


def role_prog_len(Tp, nba):
    """Chunks of one workgroup's program (csrc/role32.hpp checks it): GRU 13 Tp | E | FFN 128 tiles | tables 288 tiles."""
    kte = (4 * Tp + 31) // 32
    e = 8 if nba else 2 * kte + 8
    if nba:
        return 13 * Tp + (e + 2) // 3 + (128 + 288 + 2) // 3
    return 13 * Tp + (e + 128 + 288 + 2) // 3


def pack_posterior(sd):
    """FutureEncoder head (model/STTODE.py:258-261,297-299): out_mlp 256->128 relu, qz_layer 128->2*zdim."""
    g = lambda k: np.asarray(sd['future_encoder.' + k], np.float32)
    return {'outP': pk16(g('out_mlp.affine_layers.0.weight')), 'outb': g('out_mlp.affine_layers.0.bias'),
            'qzP': pk16(g('qz_layer.weight')), 'qzb': g('qz_layer.bias')}
