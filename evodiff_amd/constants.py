"""Alphabets of the sequence models.

Restated from ``sequence_models.constants`` (third-party, not vendored by the reference; values
cross-checked in SURVEY.md §8a) and ``evodiff/constants.py:5-8``.  Kept as plain data so a different
special-token order only needs an edit here.
"""
CAN_AAS = "ACDEFGHIKLMNPQRSTVWY"
AMB_AAS = "BZX"
OTHER_AAS = "JOU"
ALL_AAS = CAN_AAS + AMB_AAS + OTHER_AAS
GAP = "-"
STOP = "*"
MASK = "#"
START = "@"
MSA_PAD = "!"
SEP = "/"
PAD = GAP
SPECIALS = STOP + GAP + MASK + START
PROTEIN_ALPHABET = ALL_AAS + SPECIALS          # 30 symbols [recalled-upstream]: the CARP / LR-AR vocabulary
MSA_AAS = ALL_AAS + GAP
MSA_ALPHABET = ALL_AAS + GAP + STOP + MASK + START + MSA_PAD
AAINDEX_ALPHABET = "ARNDCQEGHILKMFPSTWYV"

# evodiff/constants.py: order of the BLOSUM file's rows/columns, for mapping onto MSA_ALPHABET
BLOSUM_AAS = AAINDEX_ALPHABET + AMB_AAS
BLOSUM_ALPHABET = BLOSUM_AAS + OTHER_AAS + GAP + MSA_PAD + STOP + MASK + START
BLOSUM_EXTRAS = "JOU-"
