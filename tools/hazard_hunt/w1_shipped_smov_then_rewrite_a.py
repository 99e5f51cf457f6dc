# edit of the SHIPPED (round-2) build, class-only loop of render_kernel<true>: scalar move between the MFMAs, then the A registers
# rewritten onto themselves (two vector moves) before the second MFMA reads them
edits = [("render_kernelILb1ELb0E",
          r"(\tv_mfma_f32_32x32x16_f16 v\[0:15\], v\[46:49\], v\[32:35\], 0\n)(\tv_mfma_f32_32x32x16_f16 v\[16:31\], v\[46:49\], v\[36:39\], 0\n)",
          "\\1\ts_mov_b32 s67, s67\n\tv_mov_b64_e32 v[46:47], v[46:47]\n\tv_mov_b64_e32 v[48:49], v[48:49]\n\ts_nop 1\n\\2", 1)]
