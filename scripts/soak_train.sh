#!/bin/bash
# soak: the train entry point at the headline configuration (base_c 48, 512x512, batch 8) on synthetic phantoms:
# 3 epochs x 60 batches with validation; the validation Dice must rise and nothing may go NaN
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 800 python -m att_aspp_unet_amd train --synthetic_batches 60 --epochs 3 --batch_size 8 --base_c 48 --img_size 512 --output_dir $O/ckpt > $O/train.log 2>&1; echo "train rc=$?"
grep -i "epoch\|dice\|nan\|error" $O/train.log | tail -12
rm -rf $O/ckpt
