// Generated list: exported names of the four twice-compiled sources, renamed in the -DDX_F16 build (see dx_common.h).
#pragma once
#define dx_accent_sum dx_accent_sum_f16
#define dx_add_pos dx_add_pos_f16
#define dx_attention_bwd dx_attention_bwd_f16
#define dx_attention_fwd dx_attention_fwd_f16
#define dx_channel_affine dx_channel_affine_f16
#define dx_colsum dx_colsum_f16
#define dx_condition_prosody dx_condition_prosody_f16
#define dx_conv_gemm dx_conv_gemm_f16
#define dx_conv_wgrad dx_conv_wgrad_f16
#define dx_cross_entropy dx_cross_entropy_f16
#define dx_embedding_bwd dx_embedding_bwd_f16
#define dx_ff_pair dx_ff_pair_f16
#define dx_gather_speaker_rows dx_gather_speaker_rows_f16
#define dx_l2_normalize dx_l2_normalize_f16
#define dx_ln_bwd dx_ln_bwd_f16
#define dx_ln_fwd dx_ln_fwd_f16
#define dx_mask_rows dx_mask_rows_f16
#define dx_mean_pool dx_mean_pool_f16
#define dx_mean_pool_bwd dx_mean_pool_bwd_f16
#define dx_pack_dims dx_pack_dims_f16
#define dx_pack_weights dx_pack_weights_f16
#define dx_pack_weights_batched dx_pack_weights_batched_f16
#define dx_relu_bwd dx_relu_bwd_f16
#define dx_scalar_conv_wgrad dx_scalar_conv_wgrad_f16
#define dx_transpose dx_transpose_f16
#define dx_unpack_wgrad dx_unpack_wgrad_f16
