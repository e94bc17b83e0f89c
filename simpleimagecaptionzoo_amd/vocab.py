"""Caption_Vocabulary (ClassRepository/CaptionVocabClass.py:1-19): id <-> word, unknown words -> <unk>.
Specials are ids 0..3 = <pad>, <sta>, <end>, <unk> (PreProcess/Build_caption_vocab.py:37-40)."""


class Caption_Vocabulary(object):
    def __init__(self):
        self.word2ix = {}
        self.ix2word = {}
        self.idx = 0

    def add_word(self, new_word):
        if new_word not in self.word2ix:
            self.word2ix[new_word] = self.idx
            self.ix2word[self.idx] = new_word
            self.idx += 1

    def __len__(self):
        return len(self.word2ix)

    def __call__(self, word):
        if word not in self.word2ix:
            return self.word2ix["<unk>"]
        return self.word2ix[word]


def synthetic_vocab(V):
    v = Caption_Vocabulary()
    for w in ("<pad>", "<sta>", "<end>", "<unk>"):
        v.add_word(w)
    for i in range(V - 4):
        v.add_word("w%d" % i)
    return v
