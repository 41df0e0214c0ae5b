"""Builds tests/golden/vocab.txt — a small synthetic uncased WordPiece vocabulary with BERT's
special-token ids ([PAD]=0, [UNK]=100, [CLS]=101, [SEP]=102, [MASK]=103).  Hub vocabularies are
unreachable offline; this one exists so the tokenizer / collator parity tests have a vocab.txt."""
import os
import string

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORDS = """test sentence pt should schedule consult with radiology no heavy lifting wks patient follow up
appointment dr clinic please call if you have fever chills pain chest shortness of breath take medication as
prescribed return to the emergency room for any new or worsening symptoms labs will be checked in one week
mri ct scan results pending at time discharge primary care physician within days continue home medications
stop taking aspirin start mg daily twice blood pressure check weight every morning doc start end and is was
are on by from this that it your an not may can must do need needs needed after before during until then
cardiology surgery wound incision dressing change keep dry clean shower bath drive driving weeks months
instructions activity diet regular low sodium fluid restriction liters per day oxygen liter nasal cannula
physical therapy occupational visiting nurse services arranged hospital admission history present illness
year old male female who presented found had underwent tolerated procedure well without complications"""
PIECES = """##s ##ed ##ing ##ly ##tion ##ment ##al ##ic ##ous ##er ##est ##y ##cin ##com ##my ##ks ##ol ##ine ##ide
##ate ##pril ##statin ##azole ##mab ##itis ##osis ##emia ##pathy ##ectomy ##scopy ##gram ##graphy van met lis
ator sim pan om pred war far hep ins cef az vanc gent levo cipro ##floxacin ##mycin ##cillin ##parin"""


def main():
    vocab = ["[PAD]"] + [f"[unused{i}]" for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    seen = set(vocab)

    def add(t):
        if t not in seen:
            seen.add(t)
            vocab.append(t)

    for ch in string.ascii_lowercase + string.digits + string.punctuation:
        add(ch)
    for ch in string.ascii_lowercase + string.digits:
        add("##" + ch)
    for w in WORDS.split():
        add(w)
    for p in PIECES.split():
        add(p)
    # a block of synthetic filler words w000..w399 so long synthetic sentences have in-vocab words
    for i in range(400):
        add(f"w{i:03d}")
    with open(os.path.join(ROOT, "tests", "golden", "vocab.txt"), "w") as f:
        f.write("\n".join(vocab) + "\n")
    print(len(vocab), "entries")


if __name__ == "__main__":
    main()
