cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python tools/c4_profile.py train /tmp/c4_call.pt ${C4_ITERS:-5000} 2>&1 | grep -v amdgpu.ids > gpurun_out/c4p.log
python tools/c4_profile.py run /tmp/c4_call.pt 50 2>&1 | grep -v amdgpu.ids >> gpurun_out/c4p.log
rm -rf gpurun_out/c4trace; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c4trace -o run -- python tools/c4_profile.py run /tmp/c4_call.pt 30 > /dev/null 2>&1
python tools/ktrace_sum.py gpurun_out/c4trace >> gpurun_out/c4p.log 2>&1
rm -f gpurun_out/c4trace/*kernel_trace.csv gpurun_out/c4trace/*/*kernel_trace.csv
cat gpurun_out/c4p.log
