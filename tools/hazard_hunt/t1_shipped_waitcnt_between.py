# edit of the SHIPPED build, class-only loop of render_kernel<true>: an s_waitcnt (sequencer, not scalar ALU) between the MFMAs
edits = [("render_kernelILb1ELb0E",
          r"(\tv_mfma_f32_32x32x16_f16 v\[0:15\], v\[46:49\], v\[32:35\], 0\n)(\tv_mfma_f32_32x32x16_f16 v\[16:31\], v\[46:49\], v\[36:39\], 0\n)",
          "\\1\ts_waitcnt lgkmcnt(0)\n\\2", 1)]
