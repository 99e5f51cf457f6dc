# edits of the SHIPPED build, class-only loop of render_kernel<true>
edits = [('render_kernelILb1ELb0E', '(\\ts_cmp_eq_u32 s67, s8\\n)(\\ts_waitcnt lgkmcnt\\(1\\)\\n)(\\tv_mfma_f32_32x32x16_f16 v\\[0:15\\], v\\[46:49\\], v\\[32:35\\], 0\\n)(\\tv_mfma_f32_32x32x16_f16 v\\[16:31\\], v\\[46:49\\], v\\[36:39\\], 0\\n)', '\\1\\2\\3\ts_nop 7\n\\4', 1)]
