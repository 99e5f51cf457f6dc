# edit of the SHIPPED build (control for m1): every valid sphere of a class word is a candidate
edits = [("render_kernelILb1ELb0E", r"(\tv_and_or_b32 v0, v0, v41, v40\n)", "\\1\tv_or_b32_e32 v0, v0, v41\n", 1)]
