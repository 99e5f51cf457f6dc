# edit of the FAILING build, class-only loop of render_kernel<true>: every valid sphere of the word becomes a candidate
# (the filter's result no longer matters; the exact stage decides alone)
edits = [("render_kernelILb1ELb0E", r"(\tv_and_or_b32 v0, v0, v49, v48\n)", "\\1\tv_or_b32_e32 v0, v0, v49\n", 1)]
