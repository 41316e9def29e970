// Conv2D MTL baselines: inference forward of
//   get_Doukhan_MTL_model     /root/reference/lib/proposed_architectures.py:425-511
//   get_Papakostas_MTL_model  /root/reference/lib/proposed_architectures.py:516-588
//   get_Jang_MTL_model        /root/reference/lib/proposed_architectures.py:650-764 (mel_scale_layer :622-646)
// with the MTL heads of :25-80, behind the C ABI of include/smh.h (smh_cnn_*).
//
// Design (gfx950): every Conv2D and Dense is ONE implicit-GEMM kernel on the f32 matrix cores
// (v_mfma_f32_32x32x2f32): rows = output pixels (or samples), columns = output channels, K = kh*kw*Cin.  Keras
// layouts are GEMM-ready as stored -- NHWC activations, HWIO kernels = row-major [K][Cout], Dense (in, out) =
// [K][Cout], Flatten = the NHWC image itself -- so weights are used in place, no packing pass.  The im2col gather is
// done on the fly into LDS through a per-layer table k -> (input offset, dy, dx); bias and the folded inference
// BatchNorm (y = acc*scale + shift) and the activation ride in the epilogue.  Long-K / few-row GEMMs (the Dense
// layers at small batch) are split along K into ordered partial sums (deterministic, no atomics).  Max-pooling, local
// response normalisation, Jang's mel-scale kernels (240 small strided convolutions) and the heads are small
// HBM-bound VALU kernels.
#include "smh_cnn_impl.h"

extern "C" int smh_cnn_create(const smh_cnn_cfg *cfg, smh_cnn **out) {
    SMH_REQUIRE(cfg && out, "smh_cnn_create: null argument");
    SMH_REQUIRE(cfg->n_classes == 3 || cfg->n_classes == 5, "n_classes must be 3 or 5 (got %d)", cfg->n_classes);
    SMH_REQUIRE(cfg->in_h >= 1 && cfg->in_w >= 1, "smh_cnn_create: bad input shape");
    SMH_REQUIRE(smh_device_count() > 0, "no HIP device visible: libsmh has no CPU path");
    smh_cnn *m = new smh_cnn();
    m->cfg = *cfg;
    std::vector<MelCl> mel;
    int rc = build_graph(m, mel);
    if (rc) {
        delete m;
        return rc;
    }
    // im2col tables, folded-epilogue offsets, largest activation
    std::vector<int2> lut;
    m->max_act = (size_t)cfg->in_h * cfg->in_w;
    for (Layer &L : m->layers) {
        m->max_act = std::max(m->max_act, (size_t)L.OH * L.OW * L.OC);
        if (L.op != kConv) continue;
        L.lut_off = lut.size();
        for (int k = 0; k < L.Kp; ++k) {
            if (k < L.K) {
                const int c = k % L.C, ij = k / L.C, j = ij % L.kw, i = ij / L.kw;
                lut.push_back(int2{(i * L.W + j) * L.C + c, (i & 0xffff) | (j << 16)});
            } else {
                lut.push_back(int2{0, 0x7fff | (0x7fff << 16)});  // out of every image: contributes zero
            }
        }
        L.es_off = m->n_fold;
        m->n_fold += 2 * (size_t)L.OC;
        L.wbf_off = m->n_wbf;
        m->n_wbf += (size_t)L.OC * L.Kp;
    }
    hipError_t e = hipMalloc((void **)&m->d_flat, m->n_params * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_fold, std::max<size_t>(m->n_fold, 1) * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void **)&m->d_lut, std::max<size_t>(lut.size(), 1) * sizeof(int2));
    if (e == hipSuccess && !mel.empty()) e = hipMalloc((void **)&m->d_mel, mel.size() * sizeof(MelCl));
    if (e == hipSuccess) e = hipMemcpy(m->d_lut, lut.data(), lut.size() * sizeof(int2), hipMemcpyHostToDevice);
    if (e == hipSuccess && !mel.empty()) e = hipMemcpy(m->d_mel, mel.data(), mel.size() * sizeof(MelCl), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(m->d_flat, 0, m->n_params * sizeof(float));
    if (e != hipSuccess) {
        smh_cnn_destroy(m);
        return smh::set_error(SMH_E_HIP, "smh_cnn_create: device allocation failed: %s", hipGetErrorString(e));
    }
    m->h_flat.assign(m->n_params, 0.f);
    *out = m;
    return SMH_OK;
}

extern "C" void smh_cnn_destroy(smh_cnn *m) {
    if (!m) return;
    (void)hipFree(m->d_flat);
    (void)hipFree(m->d_fold);
    (void)hipFree(m->d_lut);
    (void)hipFree(m->d_mel);
    (void)hipFree(m->d_wbf);
    delete m;
}

extern "C" size_t smh_cnn_num_params(const smh_cnn *m) { return m ? m->n_params : 0; }
extern "C" int smh_cnn_out_dim(const smh_cnn *m) { return m ? m->out_dim : SMH_E_INVALID; }
extern "C" int smh_cnn_feat_dim(const smh_cnn *m) { return m ? m->feat_dim : SMH_E_INVALID; }
extern "C" int smh_cnn_num_tensors(const smh_cnn *m) { return m ? (int)m->tensors.size() : SMH_E_INVALID; }

extern "C" int smh_cnn_tensor_info(const smh_cnn *m, int i, char *name, int name_cap, int *shape4, int *ndim,
                                   size_t *offset) {
    SMH_REQUIRE(m, "smh_cnn_tensor_info: null model");
    SMH_REQUIRE(i >= 0 && i < (int)m->tensors.size(), "smh_cnn_tensor_info: index %d out of range", i);
    const Tensor &t = m->tensors[i];
    if (name && name_cap > 0) {
        std::strncpy(name, t.name.c_str(), (size_t)name_cap - 1);
        name[name_cap - 1] = 0;
    }
    if (shape4)
        for (int d = 0; d < 4; ++d) shape4[d] = t.shape[d];
    if (ndim) *ndim = t.ndim;
    if (offset) *offset = t.off;
    return SMH_OK;
}

extern "C" int smh_cnn_set_weights(smh_cnn *m, const float *h, size_t n, void *stream) {
    SMH_REQUIRE(m && h, "smh_cnn_set_weights: null argument");
    SMH_REQUIRE(n == m->n_params, "smh_cnn_set_weights: got %zu floats, model has %zu", n, m->n_params);
    hipStream_t st = (hipStream_t)stream;
    m->h_flat.assign(h, h + n);
    // fold bias + inference BatchNorm into y = acc * scale + shift
    std::vector<float> fold(m->n_fold);
    for (const Layer &L : m->layers) {
        if (L.op != kConv) continue;
        float *es = fold.data() + L.es_off, *eb = es + L.OC;
        const float *bias = L.t_bias >= 0 ? h + m->tensors[L.t_bias].off : nullptr;
        for (int c = 0; c < L.OC; ++c) {
            float s = 1.f, t = 0.f;
            if (L.t_bn >= 0) {
                const float *g = h + m->tensors[L.t_bn].off;
                const float gamma = g[c], beta = g[L.OC + c], mean = g[2 * L.OC + c], var = g[3 * L.OC + c];
                s = gamma / std::sqrt(var + kBnEps);
                t = beta - mean * s;
            }
            es[c] = s;
            eb[c] = (bias ? bias[c] * s : 0.f) + t;
        }
    }
    m->wbf_valid = false;
    SMH_CHECK_HIP(hipMemcpyAsync(m->d_flat, h, n * sizeof(float), hipMemcpyHostToDevice, st));
    if (m->n_fold) SMH_CHECK_HIP(hipMemcpyAsync(m->d_fold, fold.data(), m->n_fold * sizeof(float), hipMemcpyHostToDevice, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));  // host staging buffers die here
    return SMH_OK;
}

extern "C" int smh_cnn_get_weights(const smh_cnn *m, float *h, size_t n, void *stream) {
    SMH_REQUIRE(m && h, "smh_cnn_get_weights: null argument");
    SMH_REQUIRE(n == m->n_params, "smh_cnn_get_weights: got room for %zu floats, model has %zu", n, m->n_params);
    // the device copy is the master: a trainer (smh_cnn_train.hip) updates it in place
    hipStream_t st = (hipStream_t)stream;
    SMH_CHECK_HIP(hipMemcpyAsync(h, m->d_flat, n * sizeof(float), hipMemcpyDeviceToHost, st));
    SMH_CHECK_HIP(hipStreamSynchronize(st));
    return SMH_OK;
}

extern "C" size_t smh_cnn_workspace_bytes(const smh_cnn *m, int N) {
    if (!m || N < 0) return 0;
    return carve(m, nullptr, N).bytes;
}

static int forward_impl(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                        size_t work_bytes, void *stream, bool bf16);

extern "C" int smh_cnn_forward_f32(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                                   size_t work_bytes, void *stream) {
    return forward_impl(m, d_x, N, d_out, d_feat, d_work, work_bytes, stream, false);
}

extern "C" int smh_cnn_forward_bf16(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                                    size_t work_bytes, void *stream) {
    return forward_impl(m, d_x, N, d_out, d_feat, d_work, work_bytes, stream, true);
}

static int forward_impl(const smh_cnn *m, const float *d_x, int N, float *d_out, float *d_feat, void *d_work,
                        size_t work_bytes, void *stream, bool bf16) {
    SMH_REQUIRE(m, "smh_cnn_forward_f32: null model");
    SMH_REQUIRE(N >= 0, "smh_cnn_forward_f32: negative batch");
    if (N == 0) return SMH_OK;
    SMH_REQUIRE(d_x && d_out && d_work, "smh_cnn_forward_f32: null argument");
    const Work w = carve(m, d_work, N);
    if (work_bytes < w.bytes)
        return smh::set_error(SMH_E_WORKSPACE, "smh_cnn_forward_f32: workspace too small (%zu < %zu bytes)", work_bytes, w.bytes);
    hipStream_t st = (hipStream_t)stream;
    if (bf16 && !m->wbf_valid) {  // bf16 operand cache: every kernel transposed to [Cout][Kp], once per weight version
        if (!m->d_wbf) SMH_CHECK_HIP(hipMalloc(&m->d_wbf, std::max<size_t>(m->n_wbf, 1) * sizeof(__bf16)));
        for (const Layer &L : m->layers)
            if (L.op == kConv) {
                const size_t tot = (size_t)L.OC * L.Kp;
                hipLaunchKernelGGL(pack_wt_bf16_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st,
                                   (const float *)(m->d_flat + m->tensors[L.t_kernel].off), L.K, L.Kp, L.OC,
                                   reinterpret_cast<__bf16 *>(m->d_wbf) + L.wbf_off);
            }
        int rcp = smh::launch_status("pack_wt_bf16_kernel");
        if (rcp) return rcp;
        m->wbf_valid = true;
    }
    const size_t in_floats = (size_t)m->cfg.in_h * m->cfg.in_w;
    for (int n0 = 0; n0 < N; n0 += kChunk) {
        const int n = std::min(kChunk, N - n0);
        const float *src = d_x + (size_t)n0 * in_floats;
        int pp = 0;
        for (const Layer &L : m->layers) {
            float *dst = w.act[pp];
            if (L.op == kConv) {
                ConvArgs a{};
                a.x = src, a.w = m->d_flat + m->tensors[L.t_kernel].off;
                a.es = m->d_fold + L.es_off, a.eb = a.es + L.OC;
                a.y = dst, a.partial = w.partial, a.lut = m->d_lut + L.lut_off;
                a.H = L.H, a.W = L.W, a.Cin = L.C, a.OH = L.OH, a.OW = L.OW, a.Cout = L.OC, a.K = L.K;
                a.M = n * L.OH * L.OW;
                a.sh = L.sh, a.sw = L.sw, a.pt = L.pt, a.pl = L.pl, a.act = L.act;
                a.ksteps = L.Kp / (bf16 ? BKB : BK);
                a.Kp = L.Kp;
                a.vec4 = (L.C % 4 == 0) ? 1 : 0;
                const int bn = L.OC <= 64 ? 64 : 128;
                const int mt = (a.M + BM - 1) / BM, nt = (L.OC + bn - 1) / bn;
                a.ksplit = choose_split(mt, nt, L.Kp / BK);  // the same plan for both precisions (workspace sizing)
                if (a.ksplit > a.ksteps) a.ksplit = a.ksteps;
                a.ksteps_per = (a.ksteps + a.ksplit - 1) / a.ksplit;
                SMH_REQUIRE(L.OC % 4 == 0, "smh_cnn: Cout=%d is not a multiple of 4", L.OC);
                const dim3 grid(mt, nt, a.ksplit);
                if (bf16) {
                    const __bf16 *wt = reinterpret_cast<const __bf16 *>(m->d_wbf) + L.wbf_off;
                    if (bn == 64) hipLaunchKernelGGL(conv_gemm_bf16_kernel<64>, grid, dim3(256), 0, st, a, wt);
                    else hipLaunchKernelGGL(conv_gemm_bf16_kernel<128>, grid, dim3(256), 0, st, a, wt);
                } else if (bn == 64) hipLaunchKernelGGL(conv_gemm_kernel<64>, grid, dim3(256), 0, st, a);
                else hipLaunchKernelGGL(conv_gemm_kernel<128>, grid, dim3(256), 0, st, a);
                if (a.ksplit > 1) {
                    const size_t MN = (size_t)a.M * L.OC;
                    hipLaunchKernelGGL(splitk_epilogue_kernel, dim3((unsigned)((MN + 255) / 256)), dim3(256), 0, st,
                                       (const float *)w.partial, a.ksplit, MN, L.OC, a.es, a.eb, L.act, dst);
                }
            } else if (L.op == kPool) {
                const size_t total = (size_t)n * L.OH * L.OW * L.C;
                hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, L.H, L.W,
                                   L.C, L.OH, L.OW, L.kh, L.kw, L.sh, L.sw, L.pt, L.pl, total, dst);
            } else if (L.op == kLrnRelu) {
                const size_t total = (size_t)n * L.H * L.W * L.C;
                hipLaunchKernelGGL(lrn_relu_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src, L.C, 5,
                                   1e-4f, 0.75f, total, dst);
            } else {  // kMelCl
                const size_t total = (size_t)n * L.OH * L.OW;
                hipLaunchKernelGGL(melcl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, src,
                                   (const MelCl *)m->d_mel, (const float *)m->d_flat, L.H, L.W, L.OH, L.kw, total, dst);
            }
            src = dst;
            pp ^= 1;
        }
        // src now holds the (n, feat_dim) features
        if (d_feat)
            SMH_CHECK_HIP(hipMemcpyAsync(d_feat + (size_t)n0 * m->feat_dim, src, (size_t)n * m->feat_dim * sizeof(float),
                                         hipMemcpyDeviceToDevice, st));
        HeadArgs ha{};
        const float *F = m->d_flat;
        ha.c3k = F + m->tensors[m->t_c3].off, ha.c3b = F + m->tensors[m->t_c3 + 1].off;
        for (int h = 0; h < m->n_heads; ++h) {
            const int t = m->t_head[h];
            ha.hk[h] = F + m->tensors[t].off, ha.hb[h] = F + m->tensors[t + 1].off;
            ha.hbn[h] = F + m->tensors[t + 2].off;  // gamma, beta, mean, var are contiguous
            ha.ok[h] = F + m->tensors[t + 6].off, ha.ob[h] = F + m->tensors[t + 7].off;
            ha.odim[h] = m->odim[h], ha.sigm[h] = m->sigm[h];
        }
        ha.D = m->feat_dim, ha.n_classes = m->cfg.n_classes, ha.n_heads = m->n_heads, ha.out_dim = m->out_dim;
        hipLaunchKernelGGL(heads_kernel, dim3(n), dim3(256), 0, st, src, ha, d_out + (size_t)n0 * m->out_dim);
    }
    return smh::launch_status("smh_cnn forward");
}
