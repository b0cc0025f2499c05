"""ctypes mirror of include/eegclip.h (struct layouts + prototypes).  Pure declarations, no compute."""
import ctypes as C

BIG = 1 << 62


class Dim(C.Structure):
    _fields_ = [("div", C.c_longlong), ("so", C.c_longlong), ("si", C.c_longlong)]


def dim(si, div=BIG, so=0):
    """offset(i) = (i // div) * so + (i % div) * si   (plain stride when div is omitted)."""
    return Dim(int(div), int(so), int(si))


class GemmDesc(C.Structure):
    _fields_ = [
        ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("A", C.c_void_p), ("Am", Dim), ("Ak", Dim),
        ("B", C.c_void_p), ("Bk", Dim), ("Bn", Dim),
        ("C", C.c_void_p), ("Cm", Dim), ("Cn", Dim),
        ("Cpre", C.c_void_p),
        ("bias_n", C.c_void_p), ("bias_m", C.c_void_p),
        ("R", C.c_void_p), ("Rm", Dim), ("Rn", Dim),
        ("alpha", C.c_float), ("accumulate", C.c_int), ("act", C.c_int),
        ("drop_p", C.c_float), ("seed", C.c_ulonglong), ("drop_site", C.c_uint),
        ("split_k", C.c_int),
        ("rowsum_a", C.c_void_p),
        ("precision", C.c_int),
        ("workspace", C.c_void_p), ("workspace_bytes", C.c_longlong),
    ]


class SplitItem(C.Structure):
    _fields_ = [("src", C.c_void_p), ("hi", C.c_void_p), ("lo", C.c_void_p), ("rows", C.c_int), ("cols", C.c_int), ("ld_src", C.c_longlong),
                ("ld_out", C.c_longlong), ("transpose", C.c_int), ("copy", C.c_void_p), ("ld_copy", C.c_longlong)]


class InfonceProblem(C.Structure):
    _fields_ = [("q_hi", C.c_void_p), ("q_lo", C.c_void_p), ("k_hi", C.c_void_p), ("k_lo", C.c_void_p), ("col0", C.c_int), ("weight", C.c_float),
                ("part", C.c_void_p), ("diag", C.c_void_p), ("lse", C.c_void_p), ("lse_k", C.c_void_p), ("G", C.c_void_p), ("ldg", C.c_longlong),
                ("part_k", C.c_void_p), ("diag_k", C.c_void_p), ("G_hi", C.c_void_p), ("G_lo", C.c_void_p)]


class TokenBlockDesc(C.Structure):
    _fields_ = ([("B", C.c_int), ("x", C.c_void_p), ("packed", C.c_void_p), ("bv", C.c_void_p), ("pe", C.c_void_p), ("tokens", C.c_void_p), ("ids", C.c_void_p)]
                + [(k, C.c_void_p) for k in ("bqkv", "bo", "ln1_g", "ln1_b", "b1", "b2", "ln2_g", "ln2_b", "ln3_g", "ln3_b")]
                + [(k, C.c_void_p) for k in ("h", "qkv", "r1", "mu1", "rs1", "f1", "r2", "n2", "mu2", "rs2", "n3", "mu3", "rs3")]
                + [(k, C.c_void_p) for k in ("xp", "hp", "ctxp", "n1p", "g1p")]
                + [("drop_p", C.c_float), ("eps", C.c_float), ("scale", C.c_float), ("seed", C.c_ulonglong)]
                + [(k, C.c_uint) for k in ("site_embed", "site_attn", "site_attn_out", "site_ffn_act", "site_ffn_out")]
                + [("packed_embed", C.c_void_p), ("embed_subject", C.c_void_p), ("bv_stride", C.c_longlong)]
                + [("cs_w25", C.c_void_p), ("cs_bias", C.c_void_p), ("cs_rows", C.c_void_p), ("cs_H", C.c_int)])


class TokenBlockBwdDesc(C.Structure):
    _fields_ = ([("B", C.c_int), ("packed", C.c_void_p)]
                + [(k, C.c_void_p) for k in ("dn3", "n2", "r2", "r1", "f1", "mu1", "rs1", "mu2", "rs2", "mu3", "rs3", "ln1_g", "ln2_g", "ln2_b", "ln3_g")]
                + [(k, C.c_void_p) for k in ("dr1", "dctx", "partials", "df2p", "dg1p", "da1p", "dr1p", "dqkvp")]
                + [(k, C.c_void_p) for k in ("dln3_g", "dln3_b", "dln2_g", "dln2_b", "dln1_g", "dln1_b")]
                + [("drop_p", C.c_float), ("seed", C.c_ulonglong)]
                + [(k, C.c_uint) for k in ("site_embed", "site_attn_out", "site_ffn_act", "site_ffn_out")])


class GemmPlanesDesc(C.Structure):
    _fields_ = [("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("b_hi", C.c_void_p), ("b_lo", C.c_void_p), ("lda", C.c_longlong), ("ldb", C.c_longlong),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("C", C.c_void_p), ("ldc", C.c_longlong), ("Cpre", C.c_void_p), ("ldcpre", C.c_longlong),
                ("bias", C.c_void_p), ("R", C.c_void_p), ("ldr", C.c_longlong), ("p_hi", C.c_void_p), ("p_lo", C.c_void_p), ("ldp", C.c_longlong),
                ("act", C.c_int), ("accumulate", C.c_int), ("planes_of", C.c_int)]


class HeadGemmDesc(C.Structure):
    _fields_ = [("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("b_hi", C.c_void_p), ("b_lo", C.c_void_p), ("lda", C.c_longlong), ("ldb", C.c_longlong),
                ("M", C.c_int), ("N", C.c_int), ("K", C.c_int), ("slices", C.c_int), ("slab_stride", C.c_longlong), ("bias", C.c_void_p),
                ("Cpre", C.c_void_p), ("ldcpre", C.c_longlong), ("act", C.c_int), ("aux", C.c_void_p), ("ldaux", C.c_longlong), ("R", C.c_void_p),
                ("ldr", C.c_longlong), ("C", C.c_void_p), ("ldc", C.c_longlong), ("p_hi", C.c_void_p), ("p_lo", C.c_void_p), ("ldp", C.c_longlong),
                ("b_kmajor", C.c_int)]


class Conv16Desc(C.Structure):
    _fields_ = [("in_", C.c_void_p), ("W", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p), ("residual", C.c_void_p),
                ("N", C.c_int), ("Hi", C.c_int), ("Wi", C.c_int), ("Cin", C.c_int), ("in_pad", C.c_int),
                ("Ho", C.c_int), ("Wo", C.c_int), ("Cout", C.c_int), ("out_pad", C.c_int),
                ("KS", C.c_int), ("stride", C.c_int), ("pad_top", C.c_int), ("pad_left", C.c_int), ("upsample", C.c_int), ("dtype", C.c_int)]


class WgradTokProblem(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("a_group_stride", C.c_longlong), ("m_groups", C.c_int), ("heads_m", C.c_int), ("heads_n", C.c_int),
                ("M", C.c_int), ("N", C.c_int), ("out", C.c_void_p), ("ldo", C.c_longlong), ("bias_out", C.c_void_p), ("bias_mfma", C.c_int),
                ("sample0", C.c_int), ("samples", C.c_int), ("sample_index", C.c_void_p)]


class CstackFwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("x", C.c_void_p), ("xs_b", C.c_longlong), ("xs_h", C.c_longlong), ("w25", C.c_void_p), ("bias1", C.c_void_p),
                ("stat1", C.c_void_p), ("nstat1", C.c_int), ("count1", C.c_double), ("eps", C.c_float), ("momentum", C.c_float),
                ("gamma1", C.c_void_p), ("beta1", C.c_void_p), ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("run_mean1", C.c_void_p),
                ("run_var1", C.c_void_p), ("nbt1", C.c_void_p), ("packed", C.c_void_p), ("bias2", C.c_void_p), ("y2", C.c_void_p), ("stat2", C.c_void_p)]


class CstackBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int), ("H", C.c_int), ("x", C.c_void_p), ("xs_b", C.c_longlong), ("xs_h", C.c_longlong), ("w25", C.c_void_p), ("bias1", C.c_void_p),
                ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("gamma1", C.c_void_p), ("beta1", C.c_void_p), ("packed_t", C.c_void_p), ("dy2", C.c_void_p),
                ("rows_out", C.c_void_p), ("stat", C.c_void_p), ("nstat", C.c_int), ("count", C.c_double), ("stat_local", C.c_void_p),
                ("nstat_local", C.c_int), ("dgamma", C.c_void_p), ("dbeta", C.c_void_p), ("dx", C.c_void_p), ("dw_partials", C.c_void_p), ("dw25", C.c_void_p)]


class WgradPlanesProblem(C.Structure):
    _fields_ = [("a_hi", C.c_void_p), ("a_lo", C.c_void_p), ("lda", C.c_longlong), ("b_hi", C.c_void_p), ("b_lo", C.c_void_p), ("ldb", C.c_longlong),
                ("rows", C.c_int), ("M", C.c_int), ("N", C.c_int), ("out", C.c_void_p), ("ldo", C.c_longlong), ("bias_out", C.c_void_p), ("slices", C.c_int)]


PLAN_MAX_ARGS = 28
PLAN_MEMSET, PLAN_JOIN, PLAN_SIDE, PLAN_SKIP, PLAN_SIDE2 = -2, -3, 1, 2, 4


class PlanArg(C.Union):
    _fields_ = [("p", C.c_void_p), ("i", C.c_longlong), ("u", C.c_ulonglong), ("d", C.c_double), ("f", C.c_float), ("i32", C.c_int), ("u32", C.c_uint)]


class PlanOp(C.Structure):
    _fields_ = [("fn", C.c_int), ("flags", C.c_int), ("a", PlanArg * PLAN_MAX_ARGS)]


ACT_NONE, ACT_GELU, ACT_SILU, ACT_GELU_GRAD = 0, 1, 2, 3
PREC_F32, PREC_BF16X3 = 0, 1
ABI_VERSION = 11
DT_BF16, DT_F16 = 0, 1
_P, _I, _F, _L, _U64, _U, _D = C.c_void_p, C.c_int, C.c_float, C.c_longlong, C.c_ulonglong, C.c_uint, C.c_double

# name -> argtypes   (restype is always int)
PROTOTYPES = {
    "eegclip_abi_version": [],
    "eegclip_gemm_f32": [C.POINTER(GemmDesc), _P],
    "eegclip_gemm_workspace_bytes": [C.POINTER(GemmDesc)],
    "eegclip_split_rows": [C.POINTER(SplitItem), _I, _P],
    "eegclip_gemm_f32_grouped": [_P, _I, _P],
    "eegclip_layernorm_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "eegclip_residual_layernorm_fwd": [_P, _P, _P, _F, _U64, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P],
    "eegclip_residual_layernorm_fwd_planes": [_P, _P, _P, _F, _U64, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _P],
    "eegclip_layernorm_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _F, _U64, _U, _P],
    "eegclip_layernorm_bwd_full_workspace_floats": [_I, _I],
    "eegclip_layernorm_bwd_full": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _F, _U64, _U, _P, _P],
    "eegclip_layernorm_bwd_params_workspace_floats": [_I, _I],
    "eegclip_layernorm_bwd_params": [_P, _P, _P, _P, _P, _P, _I, _I, _P, _P],
    "eegclip_layernorm_silu_fwd": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U, _P],
    "eegclip_silu_bwd": [_P, _P, _P, _L, _I, _F, _U64, _U, _P],
    "eegclip_timestep_embedding": [_P, _I, _I, _P, _P],
    "eegclip_ddpm_add_noise": [_P, _P, _P, _P, _P, _P, _I, _I, _P],
    "eegclip_ddpm_step": [_P, _P, _P, _F, _F, _F, _F, _F, _F, _P, _P, _P, _L, _P],
    "eegclip_prior_stage_infer": [_P, _P, _P, _P, _P, _P, _P, _I, _P, _I, _I, _F, _P],
    "eegclip_mse_loss_grad": [_P, _P, _L, _P, _P, _P],
    "eegclip_mse_loss_grad_scaled": [_P, _P, _L, _F, _P, _P, _P],
    "eegclip_bn_stats": [_P, _I, _I, _I, _P, _P],
    "eegclip_bn_finalize": [_P, _D, _F, _F, _I, _P, _P, _P, _P, _I, _P, _P],
    "eegclip_bn_finalize_rows": [_P, _I, _D, _F, _F, _I, _P, _P, _P, _P, _P, _P, _P],
    "eegclip_bn_elu_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_bn_elu_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_bn_elu_bwd_stats": [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_bn_elu_bwd_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_embed_finish": [_P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_embed_finish_bwd": [_P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_dropout_scale": [_P, _L, _F, _U64, _U, _P],
    "eegclip_gelu_bwd": [_P, _P, _P, _L, _I, _F, _U64, _U, _P],
    "eegclip_bias_act": [_P, _P, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_axpby": [_P, _P, _L, _F, _F, _P],
    "eegclip_reduce_mid": [_P, _I, _I, _I, _P, _P],
    "eegclip_colsum_blocks": [_P, _I, _I, _I, _I, _L, _P, _P],
    "eegclip_sumsq": [_P, _L, _P, _P],
    "eegclip_stage_eeg": [_P, _P, _L, _I, _I, _I, _P, _I, _I, _P],
    "eegclip_gather_rows": [_P, _L, _P, _L, _P, _I, _I, _I, _P],
    "eegclip_adamw_step": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P, _P],
    "eegclip_adamw_step_zero_grad": [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _F, _P, _P],
    "eegclip_clip_scale": [_P, _F, _P, _P],
    "eegclip_attention_fwd": [_P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U, _P],
    "eegclip_attention_bwd": [_P, _P, _P, _I, _I, _I, _I, _I, _F, _F, _U64, _U, _P],
    "eegclip_attention_bwd_x3": [_P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _U64, _U, _P],
    "eegclip_proj1x1_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P],
    "eegclip_proj1x1_fwd_rows": [_P, _P, _I, _D, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P],
    "eegclip_proj1x1_bwd_workspace_floats": [_I],
    "eegclip_proj1x1_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P],
    "eegclip_cast_bf16": [_P, _P, _L, _P],
    "eegclip_logits_bf16": [_P, _P, _P, _I, _I, _I, _L, _P, _P],
    "eegclip_tsconv_fwd_workspace_floats": [_I, _I],
    "eegclip_tsconv_fwd": [_P, _L, _L, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P],
    "eegclip_tsconv_bwd_w": [_P, _L, _L, _P, _P, _P, _I, _I, _I, _I, _P],
    "eegclip_tsconv_bwd_w_workspace_floats": [_I, _I],
    "eegclip_tsconv_bwd_x": [_P, _P, _P, _L, _L, _I, _I, _I, _I, _P],
    "eegclip_cross_attn_fwd": [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P],
    "eegclip_sconv_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _P],
    "eegclip_sconv_fwd_workspace_floats": [_I],
    "eegclip_sconv_bwd_w_workspace_floats": [_I, _I],
    "eegclip_sconv_bwd_w": [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _P],
    "eegclip_sconv_bwd_x_stats_workspace_floats": [_I],
    "eegclip_sconv_bwd_x_stats": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "eegclip_sconv_bwd_x_apply": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _D, _P, _P, _P, _I, _I, _P],
    "eegclip_lse_rows": [_P, _I, _I, _L, _P, _P, _P],
    "eegclip_lse_cols": [_P, _I, _I, _L, _P, _P, _P],
    "eegclip_infonce_grad": [_P, _I, _I, _L, _I, _I, _P, _P, _P, _F, _P, _P, _P],
    "eegclip_infonce_loss": [_P, _I, _L, _P, _P, _P, _F, _P, _P],
    "eegclip_gemm16": [_P, _L, _P, _L, _P, _L, _P, _P, _L, _I, _I, _I, _I, _I, _P],
    "eegclip_sampler_step": [_P, _P, _P, _P, _P, _P, _F, _F, _F, _F, _F, _L, _I, _P],
    "eegclip_split_bf16": [_P, _P, _P, _L, _P],
    "eegclip_infonce_fused_supported": [_I, _I, _I],
    "eegclip_infonce_fused_workspace_floats": [_I, _I],
    "eegclip_infonce_fused_fwd": [C.POINTER(InfonceProblem), _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "eegclip_infonce_fused_grad": [C.POINTER(InfonceProblem), _I, _I, _I, _I, _I, _I, _P, _P, _P],
    "eegclip_infonce_fused_grad_finalize": [C.POINTER(InfonceProblem), _I, _I, _I, _I, _I, _I, _P, _P, _P, _P],
    "eegclip_token_block_packed_bytes": [],
    "eegclip_token_block_pack": [_P, _P, _P, _P, _P, _P, _P],
    "eegclip_weight_prep": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "eegclip_token_block_packed_embed_bytes": [_I],
    "eegclip_token_block_pack_embed": [_P, _L, _I, _P, _P],
    "eegclip_token_block_fwd": [C.POINTER(TokenBlockDesc), _P],
    "eegclip_token_block_bwd_workspace_floats": [_I],
    "eegclip_token_block_bwd": [C.POINTER(TokenBlockBwdDesc), _I, _P],
    "eegclip_gemm_planes": [C.POINTER(GemmPlanesDesc), _P],
    "eegclip_split_transpose": [C.POINTER(SplitItem), _I, _P],
    "eegclip_infonce_small_supported": [_I, _I],
    "eegclip_infonce_small_workspace_floats": [_I, _I],
    "eegclip_infonce_small_fwd": [_P, _I, _L, _I, _I, _P, _P, _P],
    "eegclip_infonce_small_grad": [_I, _I, _P, _P, _F, _F, _F, _F, _P, _P, _L, _P, _P, _P],
    "eegclip_conv16": [C.POINTER(Conv16Desc), _P],
    "eegclip_groupnorm16": [_P, _I, _I, _I, _I, _I, _I, _P, _P, _F, _I, _P, _I, _P, _I, _P],
    "eegclip_softmax_rows16": [_P, _I, _I, _L, _F, _I, _P],
    "eegclip_vae_sample16": [_P, _P, _P, _L, _I, _I, _P],
    "eegclip_head_gemm_slices": [_I, _I, _I],
    "eegclip_head_gemm": [C.POINTER(HeadGemmDesc), _P],
    "eegclip_head_act": [_P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _P],
    "eegclip_head_act_bwd": [_P, _I, _L, _P, _P, _P, _P, _P, _L, _P],
    "eegclip_residual_layernorm_fwd_slabs": [_P, _P, _P, _F, _U64, _U, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _P, _P, _I, _L, _P, _P],
    "eegclip_layernorm_bwd_slabs": [_P, _I, _L, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _U64, _U, _P],
    "eegclip_proj1x1_bwd_slabs": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P],
    "eegclip_proj1x1_fwd_rows_planes": [_P, _P, _I, _D, _F, _F, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P, _P, C.POINTER(SplitItem), _I, _P],
    "eegclip_proj1x1_bwd_rows": [_P, _I, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _F, _U64, _U, _P],
    "eegclip_proj1x1_bwd_reduce": [_P, _I, _P, _P, _P, _P],
    "eegclip_bn_elu_bwd_apply_rows": [_P, _P, _P, _P, _P, _P, _P, _I, _L, _I, _D, _P, _P, _P, _I, _I, _I, _F, _U64, _U, _P],
    "eegclip_prior_stage_fwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _F, _U64, _U, _P],
    "eegclip_prior_stage_bwd": [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _F, _U64, _U, _P, _P],
    "eegclip_prior_stage_bwd_workspace_floats": [_I, _I],
    "eegclip_prior_stage_bwd_params": [_P, _I, _I, _P, _P, _P],
    "eegclip_silu_bwd_planes": [_P, _P, _P, _P, _L, _P],
    "eegclip_wgrad_tok_slices": [_I, _I],
    "eegclip_wgrad_tok_workspace_floats": [C.POINTER(WgradTokProblem), _I, _I, _I],
    "eegclip_wgrad_tok": [C.POINTER(WgradTokProblem), _I, _I, _I, _P, _I, _P],
    "eegclip_wgrad_tok_reduce": [C.POINTER(WgradTokProblem), _I, _I, _I, _P, _P],
    "eegclip_wgrad_tok_reduce_adamw": [C.POINTER(WgradTokProblem), _I, _I, _I, _P, _P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _L, _P],
    "eegclip_wgrad_planes": [C.POINTER(WgradPlanesProblem), _I, _P],
    "eegclip_tok_planes_from_f32": [_P, _L, _I, _I, _I, _I, _P, _P],
    "eegclip_cstack_packed_bytes": [_I],
    "eegclip_cstack_pack": [_P, _P, _I, _P],
    "eegclip_cstack_stats1": [_P, _L, _L, _P, _P, _P, _I, _I, _P],
    "eegclip_cstack_fwd": [C.POINTER(CstackFwdDesc), _P],
    "eegclip_cstack_packed_t_bytes": [_I],
    "eegclip_cstack_pack_t": [_P, _P, _I, _P],
    "eegclip_cstack_pack_all": [_P, _P, _P, _I, _P],
    "eegclip_cstack_bwd_stats": [C.POINTER(CstackBwdDesc), _P],
    "eegclip_cstack_bwd_workspace_floats": [_I],
    "eegclip_cstack_bwd_apply": [C.POINTER(CstackBwdDesc), _P],
    "eegclip_cstack_bwd_taps_reduce": [_P, _I, _P, _P],
    "eegclip_cstack_bwd_w2_workspace_floats": [_I, _I],
    "eegclip_cstack_bwd_w2": [_P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _P],
    "eegclip_plan_fn_id": [C.c_char_p],
    "eegclip_plan_events": [_I, C.POINTER(C.c_void_p)],
    "eegclip_plan_events_destroy": [_I, C.POINTER(C.c_void_p)],
    "eegclip_plan_run": [C.POINTER(PlanOp), _I, _I, _I, _P, _P, _P, C.POINTER(C.c_void_p), _P, _P, C.POINTER(C.c_int), C.POINTER(C.c_int)],
    "eegclip_topk_rows": [_P, _I, _I, _L, _I, _P, _P, _P],
    "eegclip_count_equal": [_P, _I, _P, _I, _P, _P],
    "eegclip_top1_count": [_P, _I, _I, _L, _P, _P, _P, _P],
    "eegclip_timing_event_create": [],
    "eegclip_timing_event_destroy": [_P],
    "eegclip_time_next_launch": [_P, _P],
    "eegclip_timing_elapsed_ms": [_P, _P],
}
RESTYPES = {"eegclip_timing_event_create": C.c_void_p, "eegclip_timing_elapsed_ms": C.c_float}


def plan_functions():
    """entry points the plan executor can dispatch: int f(..., void* stream) -- name -> argument ctypes WITHOUT the trailing stream"""
    skip = {"eegclip_plan_run", "eegclip_timing_event_destroy", "eegclip_time_next_launch", "eegclip_timing_elapsed_ms"}
    return {n: a[:-1] for n, a in PROTOTYPES.items() if a and a[-1] is _P and not n.endswith("_floats") and n not in skip}


def plan_slot(t):
    """PlanArg field for a ctypes parameter type"""
    return {C.c_int: "i32", C.c_uint: "u32", C.c_longlong: "i", C.c_ulonglong: "u", C.c_float: "f", C.c_double: "d"}.get(t, "p")


def declare(lib):
    """Attach prototypes; raises AttributeError if the library lacks a symbol the header declares."""
    for name, args in PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.restype = RESTYPES.get(name, C.c_longlong if name.endswith(("_floats", "_bytes")) else C.c_int)
        fn.argtypes = args
    return lib
