set -x
mkdir -p gpurun_out/r06
cd tools/probes && /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shape_asm mfma_shape_asm.hip 2>/dev/null; cd ../..
timeout 300 /tmp/mfma_shape_asm 2.0 3 > gpurun_out/r06/probe_mfma_shape_asm.txt 2>&1
cat gpurun_out/r06/probe_mfma_shape_asm.txt
