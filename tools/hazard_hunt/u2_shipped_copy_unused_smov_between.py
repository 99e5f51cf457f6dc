# edit of the SHIPPED build (control for u1): the same copy instructions, but the second MFMA still reads the ORIGINAL A registers; scalar move between
edits = [("render_kernelILb1ELb0E",
          r"(\ts_waitcnt lgkmcnt\(1\)\n)(\tv_mfma_f32_32x32x16_f16 v\[0:15\], v\[46:49\], v\[32:35\], 0\n)(\tv_mfma_f32_32x32x16_f16 v\[16:31\], v\[46:49\], v\[36:39\], 0\n)",
          "\\1\tv_mov_b64_e32 v[56:57], v[46:47]\n\tv_mov_b64_e32 v[58:59], v[48:49]\n\ts_nop 1\n\\2\ts_mov_b32 s67, s67\n\\3", 1)]
