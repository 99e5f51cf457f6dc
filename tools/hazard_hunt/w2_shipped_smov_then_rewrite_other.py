# control for w1: the same instructions, but the vector moves rewrite registers the MFMAs do not read (v[56:59], free in this path)
edits = [("render_kernelILb1ELb0E",
          r"(\tv_mfma_f32_32x32x16_f16 v\[0:15\], v\[46:49\], v\[32:35\], 0\n)(\tv_mfma_f32_32x32x16_f16 v\[16:31\], v\[46:49\], v\[36:39\], 0\n)",
          "\\1\ts_mov_b32 s67, s67\n\tv_mov_b64_e32 v[56:57], v[56:57]\n\tv_mov_b64_e32 v[58:59], v[58:59]\n\ts_nop 1\n\\2", 1)]
