cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python tools/probe_power_amp.py 8192 512 0.001 2>&1 | tail -2
python tools/probe_power_amp.py 8192 512 0.3 2>&1 | tail -1
rocprofv3 --kernel-trace --stats -d gpurun_out/pa_prof -o pa -- python tools/probe_power_amp.py 8192 512 0.001 > gpurun_out/pa_prof.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT -d gpurun_out/pa_pmc -o pa -- python tools/probe_power_amp.py 8192 512 0.001 > gpurun_out/pa_pmc.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE -d gpurun_out/pa_pmc2 -o pa -- python tools/probe_power_amp.py 8192 512 0.001 > gpurun_out/pa_pmc2.log 2>&1
ls gpurun_out/pa_prof gpurun_out/pa_pmc
