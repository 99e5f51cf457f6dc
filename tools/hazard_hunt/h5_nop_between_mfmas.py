# edits of the FAILING build (-DRTK_EXP_BRANCHFREE_BITMAP=1), class-only loop of render_kernel<true>
edits = [('render_kernelILb1ELb0E', '(\\tv_mfma_f32_32x32x16_f16 v\\[16:31\\], v\\[44:47\\], v\\[32:35\\], 0\\n)(\\ts_cmp_eq_u32 s4, s8\\n)(\\tv_mfma_f32_32x32x16_f16 v\\[0:15\\], v\\[44:47\\], v\\[36:39\\], 0\\n)', '\\2\\1\ts_nop 0\n\\3', 1)]
