cd $GRAFT_REPO_ROOT
echo "== C5 shard: horse subdiv (2562 v / 5120 f), 16 frames @512"; timeout -k 10 200 python tools/kbench.py --mesh horse --subdiv 1 --frames 16 --img 512 | grep -E "us/iter|sum|coverage"
echo "== C3: horse 32 @256"; timeout -k 10 100 python tools/kbench.py --mesh horse --frames 32 | grep -E "sum|coverage"
echo "== C4 shard: cow 32 @256"; timeout -k 10 100 python tools/kbench.py --mesh cow --frames 32 | grep -E "sum|coverage"
echo "== bird 16 @512"; timeout -k 10 100 python tools/kbench.py --mesh bird --frames 16 --img 512 | grep -E "sum|coverage"
echo "== bird 128 @256"; timeout -k 10 100 python tools/kbench.py --mesh bird --frames 128 | grep -E "sum|coverage"
