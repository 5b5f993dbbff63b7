# ORACLE sanitizer build (CPU only):  make -C oracle -f asan.mk
liboracle_asan.so: cdcl.c check.c oracle.h
	gcc -O1 -g -fsanitize=address,undefined -fPIC -std=c11 -D_POSIX_C_SOURCE=200809L -shared -o $@ cdcl.c check.c -lm
