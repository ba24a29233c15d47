# rocprofv3 kernel statistics of the widened rows (alias audit, render-midi); run on the GPU box from the repo root.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01w
export OW_MIDI_JOBS=512 OW_MIDI_SPAN=10 OW_MIDI_NOTES=100
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01w/midi -o t -- python3 tools/bench_midi_render.py > gpurun_out/r01w/midi.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r01w/audit -o t -- python3 tools/run_alias_audit.py > gpurun_out/r01w/audit.log 2>&1
find gpurun_out/r01w -name "*kernel_stats.csv" | head; tail -2 gpurun_out/r01w/midi.log; tail -4 gpurun_out/r01w/audit.log
