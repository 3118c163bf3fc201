# same-box A/B of a compile-time switch: tools/ab_build.sh "-DFLAG" [bench args]
flag="$1"; shift
run() { python bench.py --cpu-seconds 0 --steps 10 "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$tag', round(d['value']), round(d['ms_per_step'],3), [round(x,1) for x in r['per_level_regular_kernel_us']], round(r['first_iteration_launch_us'],1), [round(x,1) for x in r['per_level_setup_us']])"; }
for rep in 1 2; do
  tag="default"; python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1; run "$@"
  tag="$flag"; ICTR_EXTRA_HIPCC_FLAGS="$flag" python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1; run "$@"
done
python -c "import __graft_entry__ as g; g.build(force=True)" >/dev/null 2>&1
