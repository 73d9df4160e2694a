/*
 * oracle/oracle_data.c -- TEST INFRASTRUCTURE.  Embeds the two static data
 * files of the Brotli format (RFC 7932 Appendix A dictionary, Appendix B
 * transforms in this repo's 20-byte record layout, see tools/gen_tables.py)
 * into liboracle.so.  Reference counterparts: dict.h (yeast.c:896) and
 * xforms.h (yeast.c:890).
 */
__asm__(".section .rodata\n"
        ".global oracle_dictionary\n"
        ".type oracle_dictionary, @object\n"
        ".balign 16\n"
        "oracle_dictionary:\n"
        ".incbin \"" DATA_DIR "/brotli_dictionary.bin\"\n"
        ".size oracle_dictionary, 122784\n"
        ".global oracle_transforms\n"
        ".type oracle_transforms, @object\n"
        ".balign 16\n"
        "oracle_transforms:\n"
        ".incbin \"" DATA_DIR "/transforms.bin\"\n"
        ".size oracle_transforms, 2420\n"
        ".text\n");
