# edit of the SHIPPED build, class-only loop of intersect_kernel<true>: a scalar move of a register onto itself between the two MFMAs
edits = [("intersect_kernelILb1ELb0E",
          r"(\tv_mfma_f32_32x32x16_f16 v\[0:15\], v\[44:47\], v\[34:37\], 0\n)(\tv_mfma_f32_32x32x16_f16 v\[16:31\], v\[44:47\], v\[38:41\], 0\n)",
          "\\1\ts_mov_b32 s69, s69\n\\2", 1)]
