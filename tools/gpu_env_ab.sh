# usage (GPU box): bash tools/gpu_env_ab.sh <describe|orient|detect|pyramid> <ENV_NAME> <value A> <value B>  -- one frame-driver kernel of a
# 64-frame call alone (tools/ksite.py) with the environment switch at either value, alternating, five rounds
cd $GRAFT_REPO_ROOT
site=$1; name=$2; a=$3; b=$4
for i in 1 2 3 4 5; do
for v in $a $b; do
env $name=$v timeout -k 10 120 python tools/ksite.py $site 64 2>&1 | grep "^$site" | sed "s/^/$name=$v: /" || exit 1
done
done
