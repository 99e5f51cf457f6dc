# edit of the SHIPPED build
edits = [('', '(\\tv_cmp_ne_u32_e32 vcc, 0, (v\\d+)\\n)(\\tds_write_b32 (v\\d+), \\2\\n)(\\tv_add_u32_e32 \\4, 0x100, \\4\\n)\\tv_cndmask_b32_e32 \\2, 0, (v\\d+), vcc\\n', '\\1\\3\\5\\tv_cndmask_b32_e64 \\2, 0, \\6, vcc\\n', 4)]
