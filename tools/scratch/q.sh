cd "${GRAFT_REPO_ROOT:-.}"
for l in "" q4 q16; do echo "== $l"; AIPT_LIB=${l:+$PWD/ab_variants/libaiptd_$l.so} python bench.py --no-cpu-baseline --layers --steps 20 --warmup 5 2>&1 >/dev/null | grep -E "dec1.c2|conv total"; done
