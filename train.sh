#!/bin/bash
# Same interface as the reference's train.sh:  sh train.sh <dataset> <gpu-ids>   e.g.  sh train.sh coco-attngan 0,1,2
# One process per listed GPU (torch.distributed / RCCL) instead of one data_parallel process.
DATASET="$1"
GPU="${2:-0}"
NGPU=$(echo "$GPU" | awk -F, '{print NF}')
HERE="$(cd "$(dirname "$0")" && pwd)"
if [ "$DATASET" = "coco-attngan" ]; then
    echo "Starting training on the MS-COCO data set (AttnGAN + object pathway, MI355X kernels)."
    cd "$HERE/multiple-objects-gan_amd/attngan" || exit 1
    if [ "$NGPU" -gt 1 ]; then
        HIP_VISIBLE_DEVICES="$GPU" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$NGPU" \
            --master-addr 127.0.0.1 --master-port "${MASTER_PORT:-29500}" main.py --cfg cfg/coco_train.yml --gpu "$GPU" "${@:3}"
    else
        HIP_VISIBLE_DEVICES="$GPU" python main.py --cfg cfg/coco_train.yml --gpu "$GPU" "${@:3}"
    fi
elif [ "$DATASET" = "mnist" ] || [ "$DATASET" = "clevr" ] || [ "$DATASET" = "coco-stackgan-1" ] || [ "$DATASET" = "coco-stackgan-2" ]; then
    # the StackGAN-style trees: same step on the same kernels, single process.  Real data: the trees' own TextDatasets
    # (stackgan/datasets.py, round 5) read cfg.DATA_DIR; without the data sets pass  --synthetic N  after the GPU id:
    #     sh train.sh clevr 0 --synthetic 4096
    case "$DATASET" in
        mnist)           MOD=multi_mnist; CFG=mnist_train.yml;   echo "Starting training on the Multi-MNIST data set." ;;
        clevr)           MOD=clevr;       CFG=clevr_train.yml;   echo "Starting training on the CLEVR data set." ;;
        coco-stackgan-1) MOD=coco;        CFG=coco_s1_train.yml; echo "Starting training on the MS-COCO data set." ;;
        coco-stackgan-2) MOD=coco;        CFG=coco_s2_train.yml; echo "Starting training on the MS-COCO data set." ;;
    esac
    cd "$HERE" || exit 1
    HIP_VISIBLE_DEVICES="${GPU%%,*}" python -c "import mogan_loader as m; m.load(); from mogan_amd.stackgan.$MOD import main; main.main()" \
        --cfg "$HERE/multiple-objects-gan_amd/stackgan/$MOD/cfg/$CFG" --gpu "$GPU" "${@:3}"
else
    echo "Dataset argument must be either \"mnist\", \"clevr\", \"coco-stackgan-1\", \"coco-stackgan-2\", or \"coco-attngan\"."
    exit 1
fi
