for T in 32768 16384 8192; do for B in 8 16 32 64 128 256; do echo -n "target $T: "; MTG_TP_CHUNK_TARGET=$T python3 scripts/c5_one.py $B 8 2>&1 | grep config5; done; done
