"""Label set of the 7-way multilabel task, in the fixed order the classifier head emits.

Follows /root/reference/sentclf/constants.py:5-21 (LABEL_TYPES, label2abbrev).
"""

LABEL_TYPES = [
    'I-Imaging-related followup',
    'I-Appointment-related followup',
    'I-Medication-related followups',
    'I-Procedure-related followup',
    'I-Lab-related followup',
    'I-Case-specific instructions for patient',
    'I-Other helpful contextual information',
]

label2abbrev = {
    'I-Imaging-related followup': 'Imaging',
    'I-Appointment-related followup': 'Appointment',
    'I-Medication-related followups': 'Medication',
    'I-Procedure-related followup': 'Procedure',
    'I-Lab-related followup': 'Lab',
    'I-Case-specific instructions for patient': 'Patient instructions',
    'I-Other helpful contextual information': 'Other',
}

NUM_LABELS = len(LABEL_TYPES)

# special token ids of the BERT vocabularies the reference uses (bert-base-uncased,
# Bio_Discharge_Summary_BERT); a checkpoint's tokenizer may override SEP via set_sep_token_id.
PAD_ID, UNK_ID, CLS_ID, SEP_ID, MASK_ID = 0, 100, 101, 102, 103

# The reference's per-label decision thresholds (float64[7], order of LABEL_TYPES), as np.load reads
# /root/reference/clip_dnote_label_threshs.npy and clip_ttp_label_threshs.npy (test_load.py:29): data the
# reference ships, restated here so that clip_amd.precision can gate a numerics setting on them.
DNOTE_LABEL_THRESHS = (0.44, 0.31, 0.24, 0.17, 0.46, 0.35000000000000003, 0.43)
TTP_LABEL_THRESHS = (0.18, 0.6, 0.11, 0.32, 0.1, 0.1, 0.18)
