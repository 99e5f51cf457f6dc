# edits of the FAILING build (-DRTK_EXP_BRANCHFREE_BITMAP=1); see tools/hazard_hunt/README.md
edits = [('', '(\\tv_mfma[^\\n]*\\n\\t;;#ASMSTART\\n\\t;;#ASMEND\\n)\\ts_nop \\d+\\n', '\\1\\ts_nop 15\\n\\ts_nop 15\\n', None)]
