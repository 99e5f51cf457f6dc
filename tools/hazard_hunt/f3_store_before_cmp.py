# edits of the FAILING build (-DRTK_EXP_BRANCHFREE_BITMAP=1); see tools/hazard_hunt/README.md
edits = [('', '(\\tv_cmp_ne_u32_e32 vcc, 0, (v\\d+)\\n)(\\tds_write_b32 (v\\d+), \\2\\n)(\\tv_add_u32_e32 \\4, 0x100, \\4\\n)(\\tv_cndmask_b32_e64 \\2, 0, 1, vcc\\n)', '\\3\\1\\5\\6', 4)]
