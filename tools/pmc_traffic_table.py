"""Build the per-family HBM-traffic table bench.py reads (`roofline.traffic`, `traffic_source`) from a tools/pmc_summary.py summary of the
PMC passes of the bench command:   python tools/pmc_traffic_table.py profiles/r3_final_pmc_step.json 256 > profiles/r3_pmc_hbm_traffic.json
Families are bench.py's (family_of): every gemm_x3 launch, the three fused transformer-block launches, each conv kernel on its own."""
import json
import sys

FAMILIES = {
    "gemm_bf16x3": ("eeg::gemm_x3_kernel", "eeg::wgrad_tok_kernel", "eeg::wgrad_tok_reduce_kernel", "eeg::head_gemm_kernel"),
    "token_block": ("eeg::token_block_fwd_kernel", "eeg::token_block_bwd_a_kernel", "eeg::token_block_bwd_b_kernel"),
    "conv_stack": ("eeg::cstack_", "eeg::bn_finalize_rows_kernel"),
    "attention_f32_mfma": ("eeg::attention_bwd_kernel", "eeg::attention_fwd_kernel"), "attention_bf16x3": ("eeg::attention_bwd_x3_kernel",),
    "eegclip_tsconv_fwd": ("eeg::tsconv_fwd_kernel",), "eegclip_tsconv_bwd_w": ("eeg::tsconv_bwd_w_kernel",),
    "eegclip_tsconv_bwd_x": ("eeg::tsconv_bwd_x_kernel",), "eegclip_sconv_fwd": ("eeg::sconv_fwd_kernel",),
    "eegclip_sconv_bwd_w": ("eeg::sconv_bwd_w_x3_kernel", "eeg::sconv_bwd_w_kernel"),
    "eegclip_infonce_small_fwd": ("eeg::infonce_small_fwd_kernel",), "eegclip_infonce_small_grad": ("eeg::infonce_small_grad_kernel",),
    "eegclip_head_act": ("eeg::head_act_kernel",), "eegclip_head_act_bwd": ("eeg::head_act_bwd_kernel",),
    "eegclip_proj1x1_fwd_rows_planes": ("eeg::proj1x1_fwd_kernel",), "eegclip_proj1x1_bwd_rows": ("eeg::proj1x1_bwd_kernel",),
    "eegclip_bn_elu_bwd_apply_rows": ("eeg::bn_elu_bwd_apply_rows_kernel",),
    "eegclip_adamw_step": ("eeg::adamw_kernel",),
    "eegclip_sconv_bwd_x_stats": ("eeg::sconv_bwd_x_kernel<false",), "eegclip_sconv_bwd_x_apply": ("eeg::sconv_bwd_x_kernel<true",),
}


def main(path, batch):
    s = json.load(open(path))
    fams = {}
    for fam, prefixes in FAMILIES.items():
        tot, n, members, busy, cyc = 0.0, 0, [], 0.0, 0.0
        for k, v in s.items():
            if any(k.startswith(p) for p in prefixes) and "hbm_bytes_per_launch" in v:
                tot += v["hbm_bytes_per_launch"] * v["launches_sampled"]
                n += v["launches_sampled"]
                members.append(k)
                if "mfma_busy_frac" in v:                            # time-weighted over the family's kernels
                    busy += v["mfma_busy_frac"] * v["profiled_duration_us"] * v["launches_sampled"]
                    cyc += v["profiled_duration_us"] * v["launches_sampled"]
        if n:
            fams[fam] = {"hbm_bytes_per_launch": tot / n, "launches_sampled": n, "kernels": sorted(members)}
            if cyc:
                fams[fam]["mfma_busy_frac"] = round(busy / cyc, 4)
    out = {"batch": batch,
           "source": f"{path}: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python bench.py --steps 12 --warmup 3 "
                     "--no-secondary --no-cpu-baseline (tools/final_profiles.sh); hbm_bytes = (2 FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 "
                     "FETCH_SIZE correction of MI355X_MICROARCH.md), launch-weighted mean over the family's kernels (tools/pmc_summary.py, this script)",
           "families": fams}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]))
