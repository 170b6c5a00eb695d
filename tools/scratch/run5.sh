timeout 600 python tools/train_profile.py --batch 32 --math bf16 --list 90 2>&1 | tail -92
