import re
_PAT = re.compile(r"(\[[^\]]+]|Br?|Cl?|N|O|S|P|F|I|b|c|n|o|s|p|\(|\)|\.|=|#|-|\+|\\|\/|:|~|@|\?|>|\*|\$|\%[0-9]{2}|[0-9])")


def atomwise_tokenizer(smi, exclusive_tokens=None):
    return [t for t in _PAT.findall(smi)]
