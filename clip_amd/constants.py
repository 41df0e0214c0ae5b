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
