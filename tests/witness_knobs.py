"""Defaults of the witness-map kernel-shape knobs (rs_set_tuning), for tests and tools that flip and restore them."""
SUB_CT_DEFAULT = 2  # witness_sub_ct: 0 generic, 1 sub_ntt_ct_kernel, 2 sub_ntt_wide_kernel, 3 sub_ntt_wide16_kernel (4 waves per SIMD, slower)
TREE_CT_DEFAULT = 2  # witness_tree_ct: 0 generic level loop, 1 tree_columns_kernel<512, 13>, 2 tree_wide_kernel
