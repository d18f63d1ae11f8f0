for rep in 1 2; do
for it in 2048 2560 3072 3584; do echo -n "items $it: "; NDJIR_WGRAD_ITEMS=$it python bench.py --steps 20 --warmup 5 --no-extra-legs --no-cpu-baseline 2>/dev/null | tail -1 | cut -c60-95,155-190; done
done
