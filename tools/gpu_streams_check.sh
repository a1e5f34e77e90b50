# evidence for the stream / compute-pipe placement: creation-order matrix for C4, C2 timing, in-process sequences
T=${1:-r04z}
bash tools/c4_stream_matrix.sh > gpurun_out/${T}_c4_stream_matrix.txt 2>&1
bash tools/c2_stream_matrix.sh > gpurun_out/${T}_c2_stream_matrix.txt 2>&1
for q in KSF,FD,C3,C4,C4 C4,KS,C4,C3,C4,C5,C4,FD; do echo "SEQ=$q"; SEQ=$q python tools/variant_probe.py 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | cut -c1-70; done > gpurun_out/${T}_variant_sequences.txt 2>&1
