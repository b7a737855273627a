#!/bin/bash
# Runs on the GPU box (through gpurun): wall times of the four stage tools vs their CPU counterparts (tools/bench_side.py)
# and rocprofv3 kernel statistics of each tool on the same inputs.  Outputs land in gpurun_out/side/.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/side
D=/tmp/side_inputs
rm -rf $O; mkdir -p $O
cd $R && python tools/bench_side.py gen $D > $O/inputs.json 2> $O/gen.err &&
python tools/bench_side.py run $D > $O/side.json 2> $O/run.err &&
cd /tmp && export TMPDIR=/tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sort -- $R/bin/m_sort_delta < $D/sort_in.delta > /tmp/sort.out 2> $O/sort.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/maf -- $R/bin/maf_analyzer $D/analyze_in.maf > /tmp/maf.out 2> $O/maf.log &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/make -- $R/bin/mugsy_profiles make -in_maf $D/make_in.maf -out_dir $D/make_out -basename l > $O/make.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/untr -- $R/bin/mugsy_profiles untranslate -profile_paths_list $D/untr_dirs.list -in_maf $D/untr_in.maf -out_maf /tmp/untr.maf > $O/untr.log 2>&1
true
