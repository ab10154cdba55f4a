# A/B of the leaf-release rule (tools/kbench.py arms; one frame per launch, kernel time from HIP events)
set -e
A="base:lfrac=0 f8l8:lfrac=8 f8l16:lfrac=8,leaf_thresh=16 f8l24:lfrac=8,leaf_thresh=24 f8l32:lfrac=8,leaf_thresh=32 f8l64:lfrac=8,leaf_thresh=64 f6l16:lfrac=6,leaf_thresh=16 f10l16:lfrac=10,leaf_thresh=16 f12l16:lfrac=12,leaf_thresh=16 f12l32:lfrac=12,leaf_thresh=32 f4l16:lfrac=4,leaf_thresh=16 f16l64:lfrac=16,leaf_thresh=64 l16:lfrac=0,leaf_thresh=16"
for w in atrium_1080p bunny_1080p soup_1m_4k; do
  echo "== $w fused"
  timeout -k 10 300 python -u tools/kbench.py --workload $w --rounds 10 $A
done
echo "== atrium_1080p incoherent trace"
B=$(echo $A | sed 's/\(lfrac=[0-9]*\)/\1,mode=incoherent/g')
timeout -k 10 300 python -u tools/kbench.py --workload atrium_1080p --rounds 10 $B
