#!/bin/bash
# Round 5, call 2: placement probe phase 2 (what is the class a property of?), C4 with sampled histograms at other sampling ratios / region
# batches (measurement builds), the pass-C tail at C2 (1564 vs 1536 workgroups), and the default bench line with the new fields.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$1; mkdir -p $O; cd $R
( cd /tmp; timeout 120 $R/tools/probes/placement_probe > $O/probe_phase2.log 2>&1 ); cat $O/probe_phase2.log
V=$R/theia_amd/lib/variants
timeout 400 python tools/ab_plans.py --config c4 --rounds 4 --steps 10 \
  --variants "exact=;s16u12=histogram=sampled;s16u8=histogram=sampled,lib:$V/libtad_s16u8.so;s8u8=histogram=sampled,lib:$V/libtad_s8u8.so;s4u8=histogram=sampled,lib:$V/libtad_s4u8.so" > $O/ab_c4_sampled.log 2>&1
cat $O/ab_c4_sampled.log
for K in 100000 98304 100096; do
  timeout 120 python bench.py --config c2 --keys $K --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); p=d['pipeline']
print('C2 keys $K: %.4f ms/step; meta %.3f stage0 %.3f passB %.3f detect %.3f; stage0-minus-passB %.3f; frac_whole_run %.3f; cold %s' % (d['ms_per_step'], p['ms_meta'], p['ms_stage0_clear_plus_scatter'], d['roofline']['avg_kernel_ms'], p['ms_detect_and_emit'], p['ms_stage0_clear_plus_scatter']-d['roofline']['avg_kernel_ms'], d['roofline']['frac_whole_run'], json.dumps(d.get('cold'))[:400]))"
done > $O/c2_tail.log 2>&1
cat $O/c2_tail.log
timeout 600 python bench.py > $O/bench_default_line.json 2> $O/bench_default.err; tail -c 600 $O/bench_default.err
python -c "
import json; d=json.loads(open('$O/bench_default_line.json').read().strip().splitlines()[-1]); print('C2', d['ms_per_step'], d['roofline']['frac'], d['roofline']['frac_whole_run'], d.get('cold')); [print(k, v.get('ms_per_step'), v.get('roofline',{}).get('frac'), v.get('roofline',{}).get('frac_whole_run'), (v.get('cold') or {}).get('ms_first_step'), (v.get('cold') or {}).get('placement')) for k,v in d.get('other_configs',{}).items()]"
