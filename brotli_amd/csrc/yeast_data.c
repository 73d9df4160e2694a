/*
 * yeast_data.c -- embeds the two static data files of the Brotli format into
 * libyeast_mi355x.so: the RFC 7932 Appendix A dictionary (reference: dict.h,
 * included at yeast.c:896) and the Appendix B transforms in this repo's
 * 20-byte record layout (reference: xforms.h, yeast.c:890; layout described
 * in tools/gen_tables.py).
 */
__asm__(".section .rodata\n"
        ".global ys_dictionary\n"
        ".type ys_dictionary, @object\n"
        ".balign 16\n"
        "ys_dictionary:\n"
        ".incbin \"" YS_DATA_DIR "/brotli_dictionary.bin\"\n"
        ".size ys_dictionary, 122784\n"
        ".global ys_transforms\n"
        ".type ys_transforms, @object\n"
        ".balign 16\n"
        "ys_transforms:\n"
        ".incbin \"" YS_DATA_DIR "/transforms.bin\"\n"
        ".size ys_transforms, 2420\n"
        ".text\n");
